set -x
mkdir -p gpurun_out/r03o
python tools/host_profile.py 5 2>&1 | grep -v amdgpu > gpurun_out/r03o/host_profile.txt
head -90 gpurun_out/r03o/host_profile.txt
