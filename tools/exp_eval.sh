# A/B of the eval-mode MLP kernel across the timing-experiment builds: tools/exp_eval.sh "7 21 22 23" "fp16x3 fp16"
for P in $2; do
  HIP_PRECISION=$P python tools/eval_time.py
  for N in $1; do VIPNERF_HIP_LIB=$PWD/vip-nerf_amd/lib/libvipnerf_hip_exp$N.so HIP_PRECISION=$P python tools/eval_time.py; done
done
