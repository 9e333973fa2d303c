for l in libmkd_base2.so libmkd.so; do MKD_LIB_PATH=makeupdiffuse_amd/$l python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode 0 $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['kernel_classes_ms_per_eval_back_to_back']; e=d['kernel_classes_ms_per_eval']
g=sum(v for k,v in b.items() if k.startswith('gemm_')); ge=sum(v for k,v in e.items() if k.startswith('gemm_'))
print('$l', 'ms/eval', round(d['loop']['ms_per_eval'],3), '| b2b: gemm', round(g,3), 'gn', b.get('groupnorm'), 'ln', b.get('layernorm'), 'attn', b.get('attention'), 'tfm', b.get('tfm_tail'), '| per-launch-events: gemm', round(ge,3), 'gn', e.get('groupnorm'), 'ln', e.get('layernorm'))"; done
