"""MFMA-busy share of the kernels whose SQ counters tools/pmc_summary.sh collected: profiles/<tag>_*_pmc.txt -> profiles/<tag>_mfma_busy.json.
busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024): SQ_BUSY_CYCLES is summed over the 32 shader engines (it equals 32 x the
kernel's duration in shader cycles), the MFMA counter over the 1024 SIMDs (64 cycles per v_mfma_f64_16x16x4_f64).
    python3 tools/pmc_mfma_busy.py r05"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
kernels = {'synthesis': ('synthesis_pmc', 'synthesis_rot_kernel'), 'covariance': ('covprop_pmc', 'gemm_f64_kernel<MODE_COVPROP>'),
           'filters_block': ('filters_block_pmc', 'orderwise_filter_om_kernel'), 'filters_dense': ('filters_dense_pmc', 'gemm_tall_kernel (shg_dense_filter)'),
           'analysis': ('analysis_transform_pmc', 'analysis_transform_kernel'), 'analysis_operator': ('analysis_operator_pmc', 'analysis_operator_parity_kernel')}
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import source_hashes
out = {'commit': source_hashes.commit(), 'sources': {leg: source_hashes.leg_hashes(leg) for leg in ('synthesis', 'covariance', 'filters_block', 'filters_dense', 'analysis')},
       'source': 'rocprofv3 --pmc, one pass per counter group beside --kernel-trace only (tools/pmc_summary.sh), per-dispatch averages', 'kernels': {}}
for leg, (stem, kernel) in kernels.items():
    path = os.path.join(ROOT, 'profiles', '{0}_{1}.txt'.format(tag, stem))
    if not os.path.exists(path):
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r'^\s+(\w+)\s+per-dispatch\s+([0-9.e+-]+)', open(path).read(), re.M)}
    if 'SQ_BUSY_CYCLES' not in c or not c['SQ_BUSY_CYCLES']:
        continue
    cycles = c['SQ_BUSY_CYCLES'] / 32.0
    out['kernels'][leg] = {'kernel': kernel, 'file': 'profiles/{0}_{1}.txt'.format(tag, stem), 'kernel_cycles': cycles,
                           'mfma_busy': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cycles * 1024.0),
                           'insts_mfma': c.get('SQ_INSTS_MFMA'), 'insts_valu': c.get('SQ_INSTS_VALU'), 'insts_salu': c.get('SQ_INSTS_SALU'), 'insts_lds': c.get('SQ_INSTS_LDS'),
                           'wave_cycles_waiting_share': (c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES']) if c.get('SQ_WAVE_CYCLES') else None,
                           'lds_bank_conflict_share': (c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']) if c.get('SQ_LDS_IDX_ACTIVE') else None}
json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_mfma_busy.json'), 'w'), indent=1)
for leg, v in out['kernels'].items():
    print('%-14s %-40s MFMA busy %.3f' % (leg, v['kernel'], v['mfma_busy']))
