cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
P=$GRAFT_REPO_ROOT/zig_vulkan_amd/ab/libvrt_hip_prev.so; N=$GRAFT_REPO_ROOT/zig_vulkan_amd/libvrt_hip.so
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06/t_5w.log 2>&1; echo "gpu tests rc=$?"; grep -a "passed\|failed" gpurun_out/r06/t_5w.log | tail -1
run() { VRT_HIP_LIB=$1 timeout 300 python bench.py --workload $2 --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('$3 $2', round(j['value']), round(j['ms_per_step'],4), round(j['ms_per_step_single_stream'],4), {k: round(v,4) for k,v in r['kernel_ms_per_view'].items()}, r['kernel'])"; }
{ for i in 1 2; do run $P refapp_1024x576_128x64x128_b4 prev; run $N refapp_1024x576_128x64x128_b4 new; done
  for i in 1 2; do run $P refapp_1024x576_512c_b4 prev; run $N refapp_1024x576_512c_b4 new; done; } | tee gpurun_out/r06/bench_bounce5_product.txt
timeout 600 python tools/fuzz_parity.py 1500 6501 2>&1 | tail -1 | cut -c1-300
cd /tmp; timeout 600 python $GRAFT_REPO_ROOT/tools/flythrough.py refapp_1024x576_128x64x128_b4 30 2>/dev/null | tail -1 | cut -c1-900
