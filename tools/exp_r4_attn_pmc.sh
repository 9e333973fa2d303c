# round 4: what the streaming self-attention kernel waits for (PMC, one pass per counter set; kernel-trace only beside --pmc)
export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
cat > /tmp/attn_one.py <<'PY'
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from makeupdiffuse_amd import lib as mlib
lib = mlib.load(); P = lambda t: C.c_void_p(t.data_ptr())
B, T, H, dh = 8, int(os.environ.get('ATT_T', '4096')), 8, int(os.environ.get('ATT_DH', '40')); d = H * dh
q = torch.randn(B * T, d, device='cuda').bfloat16(); k = torch.randn(B * T, d, device='cuda').bfloat16(); v = torch.randn(B * T, d, device='cuda').bfloat16(); o = torch.empty_like(q)
for _ in range(3): assert lib.mkd_attention(P(q), d, P(k), d, P(v), d, P(o), d, B, T, T, H, dh, dh ** -0.5, None) == 0
torch.cuda.synchronize()
PY
pmc() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d /tmp/pa_$name -o a -- python3 /tmp/attn_one.py > /tmp/pa_$name.log 2>&1; python3 tools/summarize_prof.py $(dirname $(find /tmp/pa_$name -name 'a_counter_collection.csv' | head -1)) a --out gpurun_out/exp_r4_attn_pmc_$name.csv; grep attention gpurun_out/exp_r4_attn_pmc_$name.csv | head -3; }
pmc wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pmc inst SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
head -1 gpurun_out/exp_r4_attn_pmc_wait.csv; head -1 gpurun_out/exp_r4_attn_pmc_inst.csv; head -1 gpurun_out/exp_r4_attn_pmc_mfma.csv
