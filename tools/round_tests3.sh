# the GPU suite in the four conv modes (default = two-limb / three-product f16 kernel with atomics, fp32-MFMA fallback, three-limb / six-product form, deterministic scatter)
mkdir -p gpurun_out/r6f
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r6f/tests_default.log 2>&1 ) 2> gpurun_out/r6f/tests_default.time; tail -3 gpurun_out/r6f/tests_default.log
cp gpurun_out/parity_drift.json gpurun_out/r6f/parity_drift_default.json
( time DDK_CONV_KERNEL=1 timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/tests_kernel1.log 2>&1 ) 2> gpurun_out/r6f/tests_kernel1.time; tail -3 gpurun_out/r6f/tests_kernel1.log
( time DDK_CONV_KERNEL=3 timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/tests_kernel3.log 2>&1 ) 2> gpurun_out/r6f/tests_kernel3.time; tail -3 gpurun_out/r6f/tests_kernel3.log
cp gpurun_out/parity_drift.json gpurun_out/r6f/parity_drift_kernel3.json
( time DDK_DETERMINISTIC=1 timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/tests_det.log 2>&1 ) 2> gpurun_out/r6f/tests_det.time; tail -3 gpurun_out/r6f/tests_det.log
