# round 5: the adopted table (large list) + packed-weight replication: the tests they touch, then the default bench line
T=gpurun_out/r5f; mkdir -p $T
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_wsgemm.py tests/test_gpu_z_properties.py "tests/test_gpu_unet.py::test_full_size_frame_against_oracle" -q -p no:cacheprovider > $T/pytest_subset.log 2>&1; tail -5 $T/pytest_subset.log
timeout 600 python bench.py > $T/bench_default.json 2> $T/bench_default.err; cut -c1-600 $T/bench_default.json
