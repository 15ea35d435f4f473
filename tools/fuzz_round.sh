CS_FUZZ_KEEP_GOING=1 timeout -k 10 330 python tools/fuzz_more.py 9400 9480 mesh > gpurun_out/fuzz_mesh.log 2>&1; tail -2 gpurun_out/fuzz_mesh.log
CS_FUZZ_NATIVE=1 CS_FUZZ_KEEP_GOING=1 timeout -k 10 330 python tools/fuzz_more.py 9400 9480 mesh > gpurun_out/fuzz_mesh_native.log 2>&1; tail -2 gpurun_out/fuzz_mesh_native.log
CS_FUZZ_KEEP_GOING=1 timeout -k 10 200 python tools/fuzz_more.py 9400 9412 bigmesh > gpurun_out/fuzz_bigmesh.log 2>&1; tail -2 gpurun_out/fuzz_bigmesh.log
CS_FUZZ_KEEP_GOING=1 timeout -k 10 250 python tools/fuzz_more.py 9400 9440 sinks > gpurun_out/fuzz_sinks.log 2>&1; tail -2 gpurun_out/fuzz_sinks.log
grep -h "FAILED" gpurun_out/fuzz_mesh.log gpurun_out/fuzz_mesh_native.log gpurun_out/fuzz_bigmesh.log gpurun_out/fuzz_sinks.log | head
