set -x
mkdir -p gpurun_out/r03r
LAKO_LIB=$PWD/lako_amd/liblako_hip_exp.so python tools/lds_contention_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r03r/lds_contention.txt
cat gpurun_out/r03r/lds_contention.txt
