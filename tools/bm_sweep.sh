for bm in 128 160 192 224 256; do
  echo "BM=$bm"
  MMADA_GEMM_BM=$bm python tools/gemm_sweep.py --variants 100 --rounds 3 --m 2440,4880,9760,19520 --shapes o8:4096:512,down8:4096:1536,o4:4096:1024,down4:4096:3072,o2:4096:2048,o1:4096:4096,down2:4096:6144,qkv8:768:4096,qkv4:1536:4096,qkv2:3072:4096,gu8:3072:4096,gu4:6144:4096 2>&1 | grep -v amdgpu.ids
done
