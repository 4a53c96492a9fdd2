R=$(pwd)
for ab in 0 32 1; do
  V3D_HIP_LIB=$R/v3d_amd/lib_exp/libv3d_hip_exp.so V3D_GEMM_ABLATE=$ab timeout 120 python tools/conv_gn_bench.py --only=c3_L0_320_bare,c3_L0_320_resonly,c3_L0_320_gnout 2>&1 | grep c3_ | sed -e "s/^/ablate=$ab /" | cut -c1-40,118-190
done
