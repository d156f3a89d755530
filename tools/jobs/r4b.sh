O=gpurun_out/r4b
mkdir -p $O
for v in SWAP_W; do
OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_$v.so timeout -k 10 120 python tools/stamps_fine.py > $O/stamps_$v.txt 2>&1
echo "== $v"; grep -E "total cycles|q gemm|kv gemm|gather|mlp0|merge" $O/stamps_$v.txt
done
