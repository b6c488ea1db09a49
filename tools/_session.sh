python tools/host_api_chunks.py | tee gpurun_out/r05_host_api_chunks.json
