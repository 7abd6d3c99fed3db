cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do VOCR_WGRAD_WINO_DMA=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = j['ms_per_step_by_entry_point']
print('dma=$v', j['value'], j['ms_per_step'], 'h2d', j['h2d_inclusive']['ms_per_step'], 'poolbwd', b.get('vocr_fracpool2x2_bwd'), 'bnbwd', b.get('vocr_bn_relu_bwd'), 'wgrad', b.get('vocr_conv3x3_wgrad_wino'))"; done
