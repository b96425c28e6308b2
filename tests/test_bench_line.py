"""The driver keeps only a tail of bench.py's stdout: the LAST line must be a compact JSON object (round 3's line had
grown to 24 KB and was cut mid-object -- BENCH_r03.parsed = null).  bench.compact_line builds that line from the full
record; these tests feed it canned full records (round 3's own 24 KB record, and synthetic worst cases) and check the
size cap, the round trip and the keys the contract names."""
import copy
import json
import os

import pytest

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _canned():
    with open(os.path.join(REPO, 'profiles', 'r03', 'bench.json')) as fh:
        return json.loads(fh.read())


def _check(line, full):
    text = json.dumps(line, separators=(',', ':')) + "\n"
    assert len(text.encode()) <= bench.LINE_LIMIT, len(text)
    assert "\n" not in text[:-1]
    back = json.loads(text)
    for k in CONTRACT:
        assert k in back, k
    assert back['value'] == pytest.approx(full['value'], rel=1e-5)
    assert back['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-5)
    assert back['metric'] == full['metric'] and back['unit'] == full['unit'] and back['dtype'] == 'f64'
    assert len(back['config']['workload']) <= 200 and 'model' not in back['config']
    r = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-4)
    return back


def test_round3_record_becomes_a_line_under_the_limit():
    full = _canned()
    assert len(json.dumps(full)) > 20000          # (the record that was cut)
    back = _check(bench.compact_line(full, 'gpurun_out/bench_full.json'), full)
    cb = back['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['kind'] in ('port', 'reference') and cb['cores'] == 1
    assert cb['parity_of_timed_pass']['vectors_checked'] == 64
    assert back['roofline']['achieved_fp64']['frac'] == pytest.approx(full['roofline']['achieved_fp64']['frac'], rel=1e-5)
    assert back['roofline']['valu_issue']['frac'] == pytest.approx(full['roofline_valu_issue']['frac'], rel=1e-5)
    # the side configurations survive as a few numbers each; a byte-model fraction above 1 is never printed as THE bound
    assert back['configs']['configs4']['roofline']['bound'] == 'valu_issue'
    assert back['configs']['configs4']['roofline']['frac'] <= 1.0
    assert back['full'] == 'gpurun_out/bench_full.json'
    # no prose: no string value longer than the workload label
    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(s) for s in strings(back)) <= 200


def test_multi_gpu_record_with_eight_ranks_stays_under_the_limit():
    full = _canned()
    full['n_gpus'] = 8
    full['cpu_baseline'] = None
    full.pop('cpu_baseline_all_cores', None)
    full['ranks'] = {"world_size_seen": 8, "backend": "nccl", "nccl_version": "2.26.6",
                     "ms_per_step_by_rank": [10.123456789 + i for i in range(8)],
                     "gathered_norms_match_local_block": True}
    full['configs'] = {"configs3_sharded": {"workload": "x" * 500, "n_gpus": 8, "vectors_per_rank": [128] * 8, "ms": 3.21,
                                            "steps": 8085546.0, "value": 2.5e9, "unit": "ODE-steps/s", "scaling": "strong",
                                            "failed_vectors": 0, "gathered_norms_match_local_block_on_every_rank": True}}
    full.pop('extras', None)
    back = _check(bench.compact_line(full, None), full)
    assert back['cpu_baseline'] is None and back['ranks']['backend'] == 'nccl' and back['ranks']['world_size_seen'] == 8
    assert back['configs']['configs3_sharded']['n_gpus'] == 8


def test_oversized_optional_parts_are_shed_not_the_contract():
    full = _canned()
    # side workloads blown up far beyond anything real: the contract keys must survive, the line must fit
    for i in range(200):
        full['configs']['side_%d' % i] = copy.deepcopy(full['configs']['configs3'])
    full['ranks']['ms_per_step_by_rank'] = [1.0] * 64
    back = _check(bench.compact_line(full, 'p'), full)
    assert back['cpu_baseline']['value'] > 0


def test_emit_writes_the_whole_buffer(tmp_path, monkeypatch):
    """emit() loops over short writes (a full pipe hands os.write less than it was given)."""
    chunks = []
    real_write = os.write

    def short_write(fd, data):
        n = min(len(data), 100)
        chunks.append(bytes(data[:n]))
        return n
    monkeypatch.setattr(bench, '_REAL_STDOUT', 12345)
    monkeypatch.setattr(os, 'write', short_write)
    obj = {"k": "v" * 1000}
    bench.emit(obj)
    monkeypatch.setattr(os, 'write', real_write)
    assert json.loads(b''.join(chunks).decode()) == obj and len(chunks) > 10


def test_round4_record_and_its_committed_line_agree():
    """profiles/r04/bench_full.json (the full record of the round's last run) -> the line; the committed line is that."""
    with open(os.path.join(REPO, 'profiles', 'r04', 'bench_full.json')) as fh:
        full = json.load(fh)
    back = _check(bench.compact_line(full, 'gpurun_out/bench_full.json'), full)
    with open(os.path.join(REPO, 'profiles', 'r04', 'bench.json')) as fh:
        committed = json.loads(fh.read())
    assert committed['value'] == back['value'] and committed['roofline'] == back['roofline']
    assert back['roofline']['traffic'] is not None and back['cpu_baseline']['parity_of_timed_pass']['gpu_within_1e-8_of_tight'] is True
    assert back['configs']['configs4']['roofline']['bound'] == 'instruction_issue'
    assert back['product_default']['integrator'] in ('dop853', 'dopri45')
