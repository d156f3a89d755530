O=gpurun_out/r3z
mkdir -p $O
for L in self,cross cross,self self,self cross,cross; do
STAMPS_FINE_LAYERS=$L timeout -k 10 120 python tools/stamps_fine.py > $O/stamps_$L.txt 2>&1
echo "== $L"; grep -E "total cycles|q gemm|kv gemm|gather|mlp0" $O/stamps_$L.txt
done
