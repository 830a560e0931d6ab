# What the 8-wave bf16 weight-gradient kernel's time is made of: its ablations (diagnostics build: make -C mipnerf360_amd/csrc diag; wrong
# results for ABL != 0) at 1024 x 1024 on 524 288 rows.  ABL bits: 1 = no MFMAs, 2 = no fragment reads, 4 = no LDS-DMA, 8 = no barrier.
#   0 = the kernel | 3 = LDS-DMA + barriers alone | 6 = MFMAs + barriers alone | 5 = fragment reads + barriers alone | 14 = MFMAs alone
O=gpurun_out/r05; mkdir -p $O
for abl in 0 3 6 5 14 0; do
  echo "== M360_TN16_ABL=$abl (8-wave form)" >> $O/wgrad_bf16_ablation_8wave_form.txt
  M360_LIB=$PWD/mipnerf360_amd/libm360_diag.so M360_WGRAD_FORM=0 M360_TN16_ABL=$abl timeout 300 python tools/diag/wgrad_bf16_probe.py 2>/dev/null | grep "^{" >> $O/wgrad_bf16_ablation_8wave_form.txt
done
echo "== the one-wave form (default), no ablation" >> $O/wgrad_bf16_ablation_8wave_form.txt
M360_LIB=$PWD/mipnerf360_amd/libm360_diag.so M360_WGRAD_FORM=1 M360_TN16_ABL=0 timeout 300 python tools/diag/wgrad_bf16_probe.py 2>/dev/null | grep "^{" >> $O/wgrad_bf16_ablation_8wave_form.txt
cat $O/wgrad_bf16_ablation_8wave_form.txt
