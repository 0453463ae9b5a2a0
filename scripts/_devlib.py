"""Which library a DEVELOPMENT script binds.  The product (`suffix_amd`) loads its in-tree libsuffix_hip.so and nothing else; the
scripts here may be pointed at the development build (`make -C suffix_amd/csrc dev`: SFX_* hooks compiled in) with
SFX_DEV_LIB=suffix_amd/libsuffix_hip_dev.so — read here, in the script, never by the package."""
import os

import suffix_amd
from suffix_amd import _lib


def engine():
    path = os.environ.get("SFX_DEV_LIB")
    if path:
        _lib.set_default_engine(suffix_amd.Engine(lib_path=os.path.abspath(path)))
    return suffix_amd.default_engine()
