O=gpurun_out/r4e
mkdir -p $O
STAMPS_FINE_WAVES=1 OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_ws.so timeout -k 10 120 python tools/stamps_fine.py bf16 > $O/stamps_bf16.txt 2>&1; grep -E "total cycles|q gemm|kv gemm|gather|mlp0|merge|LN|mlp2|end|H store|per-wave" $O/stamps_bf16.txt
