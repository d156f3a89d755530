O=gpurun_out/r4d
mkdir -p $O
STAMPS_FINE_WAVES=1 OPHIP_LIB=$PWD/onepose_st_amd/lib/libonepose_hip_ws.so timeout -k 10 120 python tools/stamps_fine.py > $O/stamps.txt 2>&1; tail -8 $O/stamps.txt
