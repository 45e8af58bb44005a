set -u
cd /tmp; export TMPDIR=/tmp
R=/root/repo
python -m pytest $R/tests/test_gpu_kernels.py -q -x -k "dense_gemm" 2>&1 | tail -3
python $R/tools/bench_dense.py 2>&1 | grep -v amdgpu | tail -8
for PM in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $PM --kernel-trace -d /tmp/pmc_dense -o $PM --output-format csv -- python $R/tools/bench_dense.py > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py /tmp/pmc_dense $R/gpurun_out/dense_traffic.json $R/gpurun_out/dense_traffic.txt
grep "dense_bf3\|igemm_bf3" $R/gpurun_out/dense_traffic.txt | cut -c1-150
