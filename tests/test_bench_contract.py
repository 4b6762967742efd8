"""The bench line's contract: keys and types of the ONE JSON line `bench.py` prints, on the GPU with a short run of a small
workload (the arithmetic behind the numbers is covered elsewhere; this guards the line the driver parses), and the
`cpu_baseline` block's structure on the CPU (the oracle timed on a tiny sample)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOP_KEYS = {'metric': str, 'value': float, 'unit': str, 'n_gpus': int, 'steps': int, 'warmup': int, 'ms_per_step': float,
            'higher_is_better': bool, 'scaling': str, 'dtype': str, 'data': str, 'config': dict}
ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample')


def test_cpu_baseline_block_structure():
    """`cpu_baseline` = the oracle (kind "port") timed on the host: value / unit / cores / kind / sample, both table layouts and the
    C++ host library's streams; three blocks, the middle one reported.  (MNIST net, B = 4, 6 steps: a second of CPU time.)"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.cpu_baseline('mnist', 1000, 1.7, B=4, steps=6, warm=1)
    for k in CPU_KEYS:
        assert k in r, k
    assert r['kind'] == 'port' and r['value'] > 0 and r['cores'] >= 1 and 'oracle' in r['sample']
    blocks = r['value_of_the_three_blocks']
    assert len(blocks) == 3 and sorted(blocks)[1] == r['value']
    assert r['scalar_table_variant']['value'] > 0 and r['host_library']['randn_per_s'] > 0
    assert r['cores'] in [int(k) for k in r['threads_tried_forward_s']]


@pytest.mark.gpu
def test_bench_prints_one_line_with_the_contracts_keys():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'mnist_unet_b256_T1000', '--steps', '5', '--warmup', '2',
                        '--no-cpu-baseline', '--no-full-trajectory'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for k, typ in TOP_KEYS.items():
        assert k in j, k
        assert isinstance(j[k], (int, float) if typ is float else typ), (k, j[k])
    assert 'vs_baseline' in j and j['vs_baseline'] is None          # BASELINE.md holds no published number for this metric
    assert j['n_gpus'] == 1 and j['steps'] == 5 and j['warmup'] == 2 and j['higher_is_better'] is True and j['scaling'] == 'weak'
    assert j['data'] == 'synthetic' and j['dtype'].startswith('f32') and j['unit'] == 'samples/s'
    assert j['config']['workload'] == 'mnist_unet_b256_T1000' and 'model' not in j['config']
    assert j['value'] > 0 and j['ms_per_step'] > 0 and j['samples_finite'] is True
    roof = j['roofline']
    for k in ROOFLINE_KEYS:
        assert k in roof, k
    assert roof['bound'] in ('hbm', 'mfma') and roof['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3 and 0 < roof['frac'] < 1
    assert roof['traffic'] is None or roof['traffic'] > 0            # PMC record exists for the default workload only
    assert j.get('cpu_baseline') is None                             # switched off for this run
    # round 6: board power / shader clock sampled by rocm-smi while the timed steps (and the trajectory) run -- the keys are always
    # there; the values are null only on a box without rocm-smi (then `board.error` says why)
    for k in ('board_power_w', 'sclk_mhz', 'package_limit_w', 'board'):
        assert k in j, k
    assert isinstance(j['board'], dict) and 'windows' in j['board'] and 'timed_steps' in j['board']['windows']
    if j['board_power_w'] is not None:
        assert 0 < j['board_power_w']['mean'] <= j['board_power_w']['max'] < 3000
        assert 0 < j['sclk_mhz']['min'] <= j['sclk_mhz']['mean'] < 4000
    assert 'algorithmic_tflops_over_fp32_peak' in j and 'whole_step_frac_of_fp32_peak' not in j
    assert 'update_kernel_unfused' in j


def test_board_sampler_parses_rocm_smi_and_tolerates_its_absence():
    """BoardSampler (bench.py): the JSON of `rocm-smi --showpower --showclocks --json` as the MI355X boxes print it (profiles/r05/
    power_probe/idle.json) -> (sclk MHz, package W, limit W); a box without rocm-smi gives a block of nulls, never an exception."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    txt = ('{"card0": {"fclk clock speed:": "(1250Mhz)", "mclk clock speed:": "(2000Mhz)", "sclk clock speed:": "(1624Mhz)", "sclk clock level:": "S", '
           '"Max Graphics Package Power (W)": "1400.0", "Current Socket Graphics Package Power (W)": "1398.0"}}')
    assert bench.BoardSampler.parse(txt) == (1624, 1398.0, 1400.0)
    assert bench.BoardSampler.parse('{"card0": {}}') == (None, None, None)
    s = bench.BoardSampler(period_s=0.05)
    s.CMD = ['/nonexistent/rocm-smi-for-the-test']
    s.start()
    t0 = time.perf_counter()
    time.sleep(0.2)
    s.stop()
    blk = s.block(dict(timed_steps=(t0, time.perf_counter()), whole_trajectory=None))
    assert blk['board_power_w'] is None and blk['sclk_mhz'] is None and blk['samples'] == 0 and 'FileNotFoundError' in blk['error']
    # and with samples in hand: mean / max / min over the window, per window and over their union
    s.rows = [(1.0, 2400, 1000.0), (2.0, 1600, 1400.0), (9.0, 100, 200.0)]
    s.limit = 1400.0
    blk = s.block(dict(timed_steps=(0.5, 2.5), whole_trajectory=None))
    assert blk['board_power_w'] == dict(mean=1200.0, max=1400.0) and blk['sclk_mhz'] == dict(mean=2000, min=1600)
    assert blk['package_limit_w'] == 1400.0 and blk['windows']['timed_steps']['samples'] == 2
