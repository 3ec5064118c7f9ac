for shape in 65536:4096 131072:2048 256:4096:1024 256:16384:256 256:65536:64 256:256:16384 256:64:65536; do
  python tools/ab_variants.py $shape 0 --rounds 4 | grep -v amdgpu | sed 's/^/W4-default  /'
  TFFT_DEBUG_VARIANTS=1 TFFT_USE_DEBUG_LIB=1 TFFT_WG4_MAX_PITCH_INREGS=0 TFFT_WG4_MAX_PITCH=0 python tools/ab_variants.py $shape 0 --rounds 4 | grep -v amdgpu | sed 's/^/W8-forced   /'
done
