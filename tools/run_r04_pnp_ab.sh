# A/B of the host PnP library on one box: previous build (lib/variants/libonepose_pnp_old.so, built by hand from the previous revision) vs the tree's
bash tools/ab_bench.sh gpurun_out/r04pnp_ab20 3 20 "OPPNP_LIB=onepose_st_amd/lib/variants/libonepose_pnp_old.so" "-"
bash tools/ab_bench.sh gpurun_out/r04pnp_ab100 2 100 "OPPNP_LIB=onepose_st_amd/lib/variants/libonepose_pnp_old.so" "-"
