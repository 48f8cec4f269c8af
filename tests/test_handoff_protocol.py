"""The whole-step kernel's granule protocol as a randomised interleaving model (CPU): is re-using one granule buffer per stage, block after
block, with tag = epoch + block and an EQUALITY test in the sweep, safe?  

Model of zonos_amd/csrc/zn_step_kernel.h: A attention workgroups and S streaming workgroups run their per-block programs
    streaming:  sweep a(b) -> publish y1(b) -> sweep y1(b) -> publish x1(b) -> sweep x1(b) -> publish m(b) -> sweep m(b) -> publish x2(b)
                -> sweep x2(b) -> publish q|k|v(b)
    attention:  sweep its slice of q|k|v(b - 1) (a subset of the streaming workgroups' parts) -> publish its part of a(b)
    attention, one workgroup per key block (round 4; `kb` = (pairs, blocks per pair)):
                sweep q|k|v(b - 1) -> publish its block maximum(b) -> sweep the maxima(b) of the pair's blocks -> publish its partial(b)
                -> (the pair's block-0 workgroup only) sweep the partials(b) of the pair's blocks -> publish the pair's part of a(b)
as independent sequential processes; a scheduler picks the next process at random and lets it do ONE memory action (publish its part, or
read ONE part of the vector it is sweeping - a sweep pass is not atomic).  The hazard the kernel must never meet: a sweep that waits for
tag t reading a part that already carries a tag NEWER than t (the data of block b overwritten by block b + 1 before it was read: a
corrupted input or a wait that can never succeed).  The model checks that on every read, over many random schedules including heavily
skewed ones (a few workgroups made very slow), and that every process finishes.  Negative controls show the check has teeth: a
streaming side that sweeps only PART of every vector (so that it no longer depends on every producer), or an attention workgroup that
publishes before it has swept its q|k|v slice, does hit the hazard."""
import random

import pytest

STAGES = ["a", "y1", "x1", "m", "x2", "qkv"]


class Hazard(Exception):
    pass


def run(n_att, n_str, n_blocks, seed, slow=(), partial_sweep=None, att_publishes_early=False, max_actions=2_000_000, kb=None, combiner_publishes_early=False):
    rng = random.Random(seed)
    n_pairs, nb = kb if kb else (n_att, 1)
    if kb:
        n_att = n_pairs * nb
    tags = {st: [0] * (n_pairs if st == "a" else n_str) for st in STAGES}        # tag of each producer's part; 0 = never written
    tags["bmax"] = [0] * n_att
    tags["part"] = [0] * n_att
    qkv_subset = {i: [j for j in range(n_str) if j % max(1, n_att // 2) == i % max(1, n_att // 2)] or [0] for i in range(n_att)}

    def sweep(vec, tag, parts):
        """generator: read the parts one at a time, pass after pass, until every one carries `tag`"""
        while True:
            ok = True
            for p in parts:
                seen = tags[vec][p]
                if seen > tag:
                    raise Hazard(f"sweep of {vec} for tag {tag} read part {p} with newer tag {seen}")
                ok &= seen == tag
                yield
            if ok:
                return

    def streaming(w):
        for b in range(1, n_blocks + 1):
            chain = [("a", "y1"), ("y1", "x1"), ("x1", "m"), ("m", "x2"), ("x2", "qkv")]
            for src, dst in chain:
                parts = list(range(len(tags[src])))
                if partial_sweep in (src, "all"):
                    parts = parts[: max(1, len(parts) // 3)]                     # negative control: does not wait for every producer
                yield from sweep(src, b, parts)
                if dst == "qkv" and b == n_blocks:
                    break
                tags[dst][w] = b
                yield

    def attention(i):
        for b in range(1, n_blocks + 1):
            if att_publishes_early:
                tags["a"][i] = b                                                  # negative control: publishes before its own sweep
                yield
            if b > 1:
                yield from sweep("qkv", b - 1, qkv_subset[i])
            if not att_publishes_early:
                tags["a"][i] = b
                yield

    def attention_kb(i):
        pair, jb = i // nb, i % nb
        mine = [pair * nb + j for j in range(nb)]
        for b in range(1, n_blocks + 1):
            if b > 1:
                yield from sweep("qkv", b - 1, qkv_subset[i])
            tags["bmax"][i] = b
            yield
            yield from sweep("bmax", b, mine)
            tags["part"][i] = b
            yield
            if jb == 0:
                if combiner_publishes_early:
                    tags["a"][pair] = b                                           # negative control: the pair's output before the partials are in
                    yield
                yield from sweep("part", b, mine)
                if not combiner_publishes_early:
                    tags["a"][pair] = b
                    yield

    procs = [streaming(w) for w in range(n_str)] + [(attention_kb(i) if kb else attention(i)) for i in range(n_att)]
    weight = [0.02 if k in slow else 1.0 for k in range(len(procs))]
    alive = list(range(len(procs)))
    actions = 0
    while alive:
        k = rng.choices(alive, weights=[weight[j] for j in alive])[0]
        try:
            next(procs[k])
        except StopIteration:
            alive.remove(k)
        actions += 1
        assert actions < max_actions, "the processes do not finish (deadlock or livelock)"
    return actions


@pytest.mark.parametrize("seed", range(40))
def test_granule_reuse_is_safe_under_random_schedules(seed):
    rng = random.Random(1000 + seed)
    n_att, n_str = rng.choice([(2, 3), (4, 6), (4, 9), (8, 12)])
    procs = n_att + n_str
    slow = tuple(rng.sample(range(procs), k=rng.choice([0, 1, 2, procs // 3])))   # a few very slow workgroups: maximal skew
    run(n_att, n_str, n_blocks=6, seed=seed, slow=slow)


def test_partial_sweeps_would_break_it():
    """If the streaming workgroups swept only part of each vector they would stop depending on every producer, and a fast neighbour could
    overwrite a stage a slow one has not read: the model must find that.  (One partial stage alone is still covered by the full sweeps of
    the stages around it - the argument needs only that every block has SOME stage everybody sweeps in full.)"""
    hits = 0
    for seed in range(60):
        try:
            run(4, 9, n_blocks=6, seed=seed, slow=(0, 5), partial_sweep="all", max_actions=300_000)
        except Hazard:
            hits += 1
        except AssertionError:
            hits += 1                                                             # a wait that can never succeed shows as a livelock
    assert hits > 0


def test_attention_publishing_before_its_sweep_would_break_it():
    hits = 0
    for seed in range(60):
        try:
            run(4, 9, n_blocks=6, seed=seed, slow=(9, 10), att_publishes_early=True, max_actions=300_000)
        except (Hazard, AssertionError):
            hits += 1
    assert hits > 0


@pytest.mark.parametrize("seed", range(40))
def test_key_block_attention_exchanges_are_safe_under_random_schedules(seed):
    """Round 4's attention role: the block maxima and partials of a (row, kv head) pair travel through two more granule buffers, reused block
    after block like the others.  They sit between a workgroup's sweep of q|k|v(b - 1) and the pair's publication of a(b): when q|k|v(b)
    exists every streaming workgroup has swept a(b) of every pair, i.e. every pair's combiner has swept its partials(b), each of which was
    published after its workgroup's sweep of the maxima(b) - so no workgroup can publish maxima or partials of block b + 1 over unread ones."""
    rng = random.Random(2000 + seed)
    pairs, nb, n_str = rng.choice([(2, 2, 3), (2, 3, 5), (4, 2, 6), (3, 4, 7)])
    procs = pairs * nb + n_str
    slow = tuple(rng.sample(range(procs), k=rng.choice([0, 1, 2, procs // 3])))
    run(0, n_str, n_blocks=6, seed=seed, slow=slow, kb=(pairs, nb))


def test_a_combiner_publishing_before_its_sweep_would_break_it():
    hits = 0
    for seed in range(60):
        try:
            run(0, 6, n_blocks=6, seed=seed, slow=(7, 9), kb=(3, 3), combiner_publishes_early=True, max_actions=300_000)
        except (Hazard, AssertionError):
            hits += 1
    assert hits > 0
