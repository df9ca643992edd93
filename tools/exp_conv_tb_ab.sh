for r in 1 2 3; do for v in "0 0" "1 1"; do set -- $v
MTL_CONV_TB=$1 MTL_CONV_TB_WGRAD=$2 python bench.py --no-extras --no-cpu-baseline > /dev/null 2>&1
python - $1 <<'P'
import json,sys
d=json.load(open('gpurun_out/bench_detail.json')); f=d['roofline']['per_family']
print('TB=%s  %.2f ms/step  x3h %.3f  wgrad_sp %.3f  wgrad_x3 %.3f  gemm_x3 %.3f ms/pass (serial profile)  total %.3f' % (sys.argv[1], d['ms_per_step'], f['conv3x3_x3h_kernel']['ms_per_pass'], f['conv3x3_wgrad_sp_kernel']['ms_per_pass'], f['conv3x3_wgrad_x3_kernel']['ms_per_pass'], f['gemm_x3_kernel<.,.,.,.,3>']['ms_per_pass'], d['roofline']['serial_step']['gpu_ms_per_pass']))
P
done; done
