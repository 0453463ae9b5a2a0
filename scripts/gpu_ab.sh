run() { # lib nw kpt
  SFX_LIB=$PWD/$1 SFX_RADIX_NW=$2 SFX_RADIX_KPT=$3 timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-microbench 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 nw$2 kpt$3', d['value'], d['ms_per_step'], d['verified'], d['roofline']['kernel_ms']['radix_scatter_u32'], d['roofline']['kernel_ms']['radix_scatter_text_u32'])"
}
for rep in 1 2; do
run suffix_amd/libsuffix_hip.so 16 8
run suffix_amd/libsuffix_hip.so 16 16
run suffix_amd/libsuffix_hip.so 8 16
run lab/libs/libsuffix_hip_mw4.so 8 16
run lab/libs/libsuffix_hip_mw4.so 16 8
done
