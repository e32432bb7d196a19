cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
P=$GRAFT_REPO_ROOT/zig_vulkan_amd/ab/libvrt_hip_prev.so; D=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip_dev.so
run() { VRT_HIP_LIB=$1 timeout 300 python bench.py --workload refapp_1024x576_128x64x128_b4 --variant $2 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$3', round(j['value']), round(j['ms_per_step'],4), round(j['ms_per_step_single_stream'],4), {k: round(v,4) for k,v in r['kernel_ms_per_view'].items()}, r['kernel'])"; }
for i in 1 2; do run $P 0 "prev(125@4)"; run $D 0 "dev 107@4 "; run $D 0x500 "dev  96@5 "; done | tee gpurun_out/r06/bench_bounce_waves.txt
