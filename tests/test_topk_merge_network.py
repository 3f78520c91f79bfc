"""CPU checks of the generated merge networks of the buffered top-k selection (gkgnet_amd/csrc/gkg_topk_merge.h):
the generator's compare-exchange lists are simulated against a plain sort, and the committed header must be what the
generator emits."""
import importlib.util
import math
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_topk_merge", os.path.join(ROOT, "tools", "gen_topk_merge.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

INF = math.inf


def run_network(KD, key, batch):
    b = list(batch)
    for i, j in gen.oddeven_merge_sort(gen.NB):
        b[i], b[j] = min(b[i], b[j]), max(b[i], b[j])
    assert b == sorted(b)
    n, ops = gen.merge_network(KD)
    pos = {i: key[i] for i in range(KD)}
    for j in range(gen.NB):
        pos[n - gen.NB + j] = b[gen.NB - 1 - j]
    for kind, i, j in ops:
        if kind == "mv":
            pos[i] = pos.pop(j)
        else:
            pos[i], pos[j] = min(pos[i], pos[j]), max(pos[i], pos[j])
    return [pos[i] for i in range(KD)]


@pytest.mark.parametrize("KD", [18, 27, 36, 64])
def test_merge_network_equals_sorted_union(KD):
    rng = random.Random(100 + KD)
    for trial in range(1500):
        filled = rng.choice([0, 1, KD // 3, KD - 1, KD])
        # a coarse value grid: plenty of exact ties between the list and the batch
        key = sorted(rng.randrange(40) / 8.0 for _ in range(filled)) + [INF] * (KD - filled)
        count = rng.randint(0, gen.NB)
        batch = [rng.randrange(40) / 8.0 for _ in range(count)] + [INF] * (gen.NB - count)
        rng.shuffle(batch)                                   # empty slots anywhere (the kernel fills from slot 0; be stricter)
        assert run_network(KD, key, batch) == sorted(key + batch)[:KD], (KD, trial)


def test_sort_network_size_and_committed_header_is_current(tmp_path):
    assert len(gen.oddeven_merge_sort(16)) == 63
    body = []
    for KD in (18, 27, 36, 64):
        lines, nce = gen.emit(KD)
        body.append((KD, nce, lines))
    committed = open(os.path.join(ROOT, "gkgnet_amd", "csrc", "gkg_topk_merge.h")).read()
    for KD, nce, lines in body:
        assert f"// KD = {KD}: 63 + {nce} compare-exchanges" in committed
        assert "\n".join(lines) in committed, f"gkg_topk_merge.h is stale for KD = {KD}: run tools/gen_topk_merge.py"
