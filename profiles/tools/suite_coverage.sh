#!/bin/bash
# usage (GPU box, repo root): bash profiles/tools/suite_coverage.sh  -> gpurun_out/gpu_suite_kernels.txt
# The -m gpu suite under rocprofv3 --kernel-trace --stats (the program directly after `--`): which instantiations does it launch?
R=$PWD
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/cov
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/cov --output-format csv -- python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider > $R/gpurun_out/cov_pytest.log 2>&1
tail -3 $R/gpurun_out/cov_pytest.log
cd $R
python3 profiles/tools/suite_coverage.py gpurun_out/cov > gpurun_out/gpu_suite_kernels.txt
rm -rf gpurun_out/cov
wc -l gpurun_out/gpu_suite_kernels.txt
