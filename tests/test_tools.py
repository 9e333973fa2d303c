"""Host-side tooling that decides which kernels run: the tuned-table generator's merge rules (CPU only)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entry(shape, cfg, s, us, dus=99.0):
    return {'shape': list(shape) + [0, 0, 0, 0, 0], 'count': 1, 'best_cfg': cfg, 'best_splitk': s, 'best_us': us, 'default_us': dus, 'trials': []}


def test_tuned_table_merge_rules(tmp_path, monkeypatch):
    """Within one sweep generation the fastest measurement of a shape wins; a later generation replaces an entry only when it
    is > 3 % faster; in-eval files (measured against the then-current table in the same run) override unconditionally."""
    a, b, c = (64, 1280, 1280, 0, 0, 0), (128, 640, 640, 0, 0, 0), (256, 320, 320, 0, 0, 0)
    files = {
        'tune2_x.json': {'a': _entry(a, 5, 1, 10.0), 'b': _entry(b, 3, 1, 20.0)},
        'tune2_y.json': {'a': _entry(a, 4, 2, 9.0), 'none': dict(_entry(c, 1, 1, 5.0), best_cfg=None)},
        'tune3_z.json': {'a': _entry(a, 14, 1, 8.9), 'b': _entry(b, 16, 1, 18.0)},       # a: < 3 % better -> stays; b: replaced
        'ineval_q.json': {'a': _entry(a, 19, 4, 50.0)},                                     # slower number, still applied
    }
    paths = []
    for name, d in files.items():
        (tmp_path / name).write_text(json.dumps(d)); paths.append(str(tmp_path / name))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_tuned_table as g
    out = tmp_path / 'table.inc'
    monkeypatch.setattr(g, 'OUT', str(out))
    monkeypatch.setattr(sys, 'argv', ['gen_tuned_table.py'] + paths)
    g.main()
    rows = [l.split('//')[0].strip() for l in out.read_text().splitlines() if l.startswith('{')]
    assert rows == ['{64, 1280, 1280, 0, 0, 0, 19, 4},', '{128, 640, 640, 0, 0, 0, 16, 1},']


def test_committed_table_matches_its_sources():
    """gemm_tuned.inc is reproducible from the committed sweep files (profiles/tune/*.json)."""
    import glob
    srcs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'tune', '*.json')))
    assert srcs
    env = dict(os.environ)
    r = subprocess.run([sys.executable, '-c', (
        'import sys, os; sys.path.insert(0, %r); import gen_tuned_table as g; g.OUT = os.devnull if False else %r; '
        'sys.argv = ["x"] + %r; g.main()') % (os.path.join(ROOT, 'tools'), '/tmp/_mkd_table_check.inc', srcs)],
        capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    strip = lambda t: [l.split('//')[0].strip() for l in t.splitlines() if l.startswith('{')]
    have = strip(open(os.path.join(ROOT, 'makeupdiffuse_amd', 'csrc', 'gemm_tuned.inc')).read())
    want = strip(open('/tmp/_mkd_table_check.inc').read())
    assert have == want


def test_tuner_scripts_know_the_librarys_tile_table():
    """tools/tune_*.py carry their own copies of the tile sizes (candidate filtering, reports): they must match kTileM / kTileN of
    kernels_gemm.hip entry for entry, and treat the same configurations as LDS-staged conv tiles as is_patch_cfg does."""
    import ast
    import re
    src = open(os.path.join(ROOT, 'makeupdiffuse_amd', 'csrc', 'kernels_gemm.hip')).read()

    def c_array(name):
        body = re.search(r'static const int ' + name + r'\[N_TILE_CFG\] = \{([^}]*)\}', src).group(1)
        return [int(v) for v in body.replace('\n', ' ').split(',')]
    n = int(re.search(r'constexpr int N_TILE_CFG = (\d+);', src).group(1))
    tm, tn = c_array('kTileM'), c_array('kTileN')
    assert len(tm) == n and len(tn) == n and len(c_array('kTileKW')) == n and len(c_array('kTileLight')) == n and len(c_array('kTileBase')) == n
    names = re.search(r'kTileName\[N_TILE_CFG\] = \{(.*?)\};', src, re.S).group(1)
    assert len(re.findall(r'"[^"]+"', names)) == n
    patch_c = re.search(r'static bool is_patch_cfg\(int c\) \{ return (.*?); \}', src).group(1)
    is_patch = lambda c: eval(patch_c.replace('&&', ' and ').replace('||', ' or '), {'c': c})
    for script in ('tune_gemm.py', 'tune_ineval.py', 'tune_wall.py'):
        text = open(os.path.join(ROOT, 'tools', script)).read()
        for var, want in (('TILE_M', tm), ('TILE_N', tn)):
            got = ast.literal_eval(re.search(r'^' + var + r' = (\[.*\])$', text, re.M).group(1))
            assert got == want, f'{script}: {var} differs from the library table'
        expr = re.search(r'patch = (.*)$', text, re.M).group(1)
        for cfg in range(n):
            assert bool(eval(expr, {'cfg': cfg})) == bool(is_patch(cfg)), f'{script}: configuration {cfg} patch / gather mismatch'
