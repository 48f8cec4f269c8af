"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol include/zonos_hip.h
declares, the ctypes binding covers them all, and argument errors come back as statuses (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from zonos_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "zonos_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from zonos_amd import build
    build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in zonos_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)


def test_abi_version_and_kv_bytes(lib):
    assert lib.zn_abi_version() == _lib.ZN_ABI_VERSION
    zc = _lib.zn_config(d_model=2048, n_layer=26, n_heads=16, n_heads_kv=4, d_ff=8192, n_codebooks=9, vocab_head=1025,
                        vocab_embed=1032, eos_id=1024, mask_id=1025, rope_positions=16384, double_out_proj=1, norm_eps=1e-5)
    # 53 248 B per row-position over 26 layers
    assert lib.zn_kv_bytes_per_layer(C.byref(zc), 1, 1) * 26 == 53248


def test_bad_arguments_return_status_not_crash(lib):
    h = C.c_void_p()
    zc = _lib.zn_config(d_model=100, n_layer=1, n_heads=3, n_heads_kv=1, d_ff=64, n_codebooks=9, vocab_head=1025,
                        vocab_embed=1032, eos_id=1024, mask_id=1025, rope_positions=16384, double_out_proj=1, norm_eps=1e-5)
    w = _lib.zn_weights()
    assert lib.zn_create(C.byref(zc), C.byref(w), 2, C.byref(h)) < 0
    assert b"divide" in lib.zn_last_error(None)
    assert lib.zn_decode_steps(None, 1, None) < 0
    assert lib.zn_destroy(None) == 0


def test_persistent_kernel_tenancy_is_one_owner_per_device(lib):
    """Two handles on one device cannot both select the persistent kernels at once (their hand-offs need every workgroup of the grid
    resident): the second generation to begin gets the launches path.  Host logic only - zn_gen_begin / zn_gen_end / zn_destroy go
    through these two primitives."""
    a, b = C.c_void_p(0x1000), C.c_void_p(0x2000)
    dev = 63                                   # a device index no test engine uses
    assert lib.zn_tenant_try_claim(dev, a) == 1
    assert lib.zn_tenant_try_claim(dev, a) == 1, "re-claiming one's own device is idempotent"
    assert lib.zn_tenant_try_claim(dev, b) == 0, "a second owner must be refused while the first holds the device"
    assert lib.zn_tenant_try_claim(dev - 1, b) == 1, "tenancy is per device"
    assert lib.zn_tenant_release(dev, b) == 0, "only the owner releases"
    assert lib.zn_tenant_release(dev, a) == 1
    assert lib.zn_tenant_try_claim(dev, b) == 1
    assert lib.zn_tenant_release(dev, b) == 1 and lib.zn_tenant_release(dev - 1, b) == 1
    assert lib.zn_tenant_try_claim(64, a) == 0 and lib.zn_tenant_try_claim(-1, a) == 0 and lib.zn_tenant_try_claim(0, None) == 0


def test_handoff_timeout_policy_is_not_silent(monkeypatch, capsys):
    """`Zonos._with_timeout_policy` (the only place a voided generation is repeated): with ZONOS_HIP_NO_TIMEOUT_RETRY=1 - what
    tests/conftest.py sets for every test - a reported hand-off timeout is raised; without it the generation is repeated once, counted
    and announced; a caller that has seen frames, or any other error, always gets the exception."""
    import os
    import types
    from zonos_amd import _lib
    from zonos_amd.model import Zonos
    assert os.environ.get("ZONOS_HIP_NO_TIMEOUT_RETRY") == "1", "tests/conftest.py must disable the retry for every test"
    m = types.SimpleNamespace(_repeats=0)
    calls = []

    def run():
        calls.append(1)
        if len(calls) == 1:
            raise _lib.ZonosHipError("zn_all_stopped_end failed (status -3): decode chain: 1 hand-off wait(s) timed out - ...")
        return "codes"
    with pytest.raises(_lib.ZonosHipError, match="hand-off wait"):
        Zonos._with_timeout_policy(m, run, caller_saw_frames=False)
    assert m._repeats == 0 and len(calls) == 1
    monkeypatch.delenv("ZONOS_HIP_NO_TIMEOUT_RETRY")
    calls.clear()
    assert Zonos._with_timeout_policy(m, run, caller_saw_frames=False) == "codes"
    assert m._repeats == 1 and len(calls) == 2 and "repeating the generation" in capsys.readouterr().err
    calls.clear()
    with pytest.raises(_lib.ZonosHipError):
        Zonos._with_timeout_policy(m, run, caller_saw_frames=True)
    assert m._repeats == 1

    def other():
        raise _lib.ZonosHipError("zn_prefill failed (status -2): something else")
    with pytest.raises(_lib.ZonosHipError, match="something else"):
        Zonos._with_timeout_policy(m, other, caller_saw_frames=False)


def test_persistent_kernels_use_no_scratch_and_fit_one_workgroup_per_cu(lib, tmp_path):
    """The persistent kernels' hand-offs wait on every workgroup of the grid: all 256 must be resident at once, whatever else the
    queue's scratch pool is doing.  Build-time facts checked on the shipped code object: no private segment (no scratch-wave slots
    involved in residency), no spills, and LDS + waves of ONE workgroup fit a CU (160 KB, 8 waves of <= 256 VGPRs)."""
    import shutil
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(OBJDUMP) and os.path.exists(readelf)):
        pytest.skip("llvm tools not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    seen = {}
    for co in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([readelf, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for k in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", k).group(1)
            if name.startswith("_Z12chain_kernel") or name.startswith("_Z11step_kernel"):
                g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
                seen[name] = dict(scratch=g("private_segment_fixed_size"), spill=g("vgpr_spill_count"), vgpr=g("vgpr_count"),
                                  lds=g("group_segment_fixed_size"), threads=g("max_flat_workgroup_size"))
    assert len(seen) == 9, sorted(seen)                  # 6 chain instantiations + 3 whole-step ones (launches covering <= 6 / 8 / 12 key blocks)
    for name, r in seen.items():
        assert r["scratch"] == 0 and r["spill"] == 0, (name, r)
        waves = r["threads"] // 64
        assert waves * r["vgpr"] <= 4 * 512, (name, r)                       # the CU's four SIMDs hold 512 VGPRs per lane each
        dyn = 128 * 1024 if name.startswith("_Z11step_kernel") else 64 * 1024    # ZN_SK_DYN_LDS / ZN_CH_DYN_LDS of the launch
        assert r["lds"] + dyn <= 160 * 1024, (name, r)


# ---------------------------------------------------------------------------------------------- emitted hand-off ISA
# The cross-workgroup hand-offs (split-K combines of gemm16s_kernel / gemm64s_kernel, per-block combine of the split P.V pass, the
# persistent chain's granules) order relaxed agent-scope accesses by instruction selection, not by fences: partial results leave by
# write-through (sc1) stores, the storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's barrier and the arrival
# atomic, and every load of handed-off bytes is an sc1 load (MI355X_MICROARCH.md "Valid forms", table row 1;
# cdna_hip_programming.md Guideline 16 R1).  Nothing in the memory model promises that selection, so the built code object
# is disassembled and checked.
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def kernels_isa(lib, tmp_path_factory):
    import shutil
    import subprocess
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    tmp = tmp_path_factory.mktemp("isa")
    so = shutil.copy(_lib.LIB_PATH, tmp / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp)       # writes the bundles next to the copy
    out = {}
    for co in sorted(tmp.glob("lib.so.*gfx950*")):
        txt = subprocess.run([OBJDUMP, "-d", str(co)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                name = m.group(1)
                out[name] = []
            elif name and line.strip():
                out[name].append(line.split("//")[0].strip())
    assert out, "no gfx950 code object found in the library"
    return out


def _check_ticket_kernel(ins):
    at = [i for i, l in enumerate(ins) if l.startswith("global_atomic_add")]
    assert len(at) == 1, "one arrival atomic expected"
    a = at[0]
    before = ins[max(0, a - 60):a]
    bar = max(i for i, l in enumerate(before) if l.startswith("s_barrier"))
    drain = [i for i, l in enumerate(before[:bar]) if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
    assert drain and bar - drain[-1] <= 3, "the storing waves must drain (vmcnt(0)) right before the barrier in front of the ticket"
    stores = [l for l in before[:drain[-1]] if l.startswith("global_store")]
    assert stores and all(" sc1" in l for l in stores), f"partial stores must be write-through: {stores}"
    after = ins[a:]
    sc1_loads = [l for l in after if l.startswith("global_load_dword ") and " sc1" in l]
    assert len(sc1_loads) >= 8, "the last arriver reads the partials with sc1 loads"


def test_ticketed_combines_use_write_through_stores_and_sc1_loads(kernels_isa):
    names = [n for n in kernels_isa if re.match(r"_Z14gemm16s_kernelILi\d+ELi\d+ELb[01]EE", n) or re.match(r"_Z17attn_block_kernelILi\d+ELi\d+ELb0ELi\d+EE", n)
             or re.match(r"_Z14gemm64s_kernelILi\d+EE", n)]
    assert len(names) >= 25            # gemm16s: 5 epilogues x 2 sizes + the LayerNorm-statistics form; attn_block (split shape): head sizes x groups x column parts
    for n in names:
        _check_ticket_kernel(kernels_isa[n])


def test_chain_kernel_handoffs_are_granules_and_sc1_sweeps(kernels_isa):
    """The persistent chain publishes 8-byte {tag, data} granules with ONE write-through store each (the data is the flag:
    Guideline 16 R2) and every load of handed-off bytes is an sc1 buffer load."""
    names = [n for n in kernels_isa if n.startswith("_Z12chain_kernel") or n.startswith("_Z11step_kernel")]
    assert len(names) == 9                                                            # 6 chain instantiations + 3 whole-step ones
    for n in names:
        ins = kernels_isa[n]
        assert not any(l.startswith("scratch_") for l in ins), n
        sweeps = [l for l in ins if l.startswith("buffer_load_dwordx4")]
        if n.startswith("_Z11step_kernel"):
            # the attention workgroups' K / V prefetch reads cache rows written by EARLIER launches (the newest row comes through
            # granules): plain buffer loads, 16 (K of the block) + 16 (full-width V) per issue site, two sites (kernel start, end of a block)
            plain = [l for l in sweeps if " sc1" not in l]
            assert len(plain) == 64, (n, len(plain))
            # e sums of the partials: one 8-byte granule per block
            assert any(l.startswith("buffer_load_dwordx2") and " sc1" in l for l in ins), n
            assert all(" sc1" in l for l in ins if l.startswith("buffer_load_dwordx2")), n
            sweeps = [l for l in sweeps if " sc1" in l]
            assert len(sweeps) >= 20, n
        assert sweeps and all(" sc1" in l for l in sweeps), n
        granules = [l for l in ins if l.startswith("global_store_dwordx2") and " sc1" in l]
        assert len(granules) >= 3, n                                                  # y1, x1, m (+ x2 when the next in_proj follows)
        assert not any(l.startswith("global_atomic_add") and "sc1" in l for l in ins)


def test_sampler_tail_ticket_orders_the_token_handoff(kernels_isa):
    """sample_kernel's last workgroup runs the step's bookkeeping + next embedding: the tokens of the other workgroups leave by ONE
    write-through store, drained before the barrier and the arrival atomic, and are read back with sc1 loads."""
    ins = kernels_isa["_Z13sample_kernel10SampleArgs"]
    at = [i for i, l in enumerate(ins) if l.startswith("global_atomic_add")]
    assert len(at) == 1
    a = at[0]
    before = ins[max(0, a - 40):a]
    bar = max(i for i, l in enumerate(before) if l.startswith("s_barrier"))
    drain = [i for i, l in enumerate(before[:bar]) if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
    assert drain and bar - drain[-1] <= 3
    stores = [l for l in before[:drain[-1]] if l.startswith("global_store")]
    assert stores and " sc1" in stores[-1], stores
    sc1_loads = [l for l in ins[a:] if l.startswith("global_load_dword ") and " sc1" in l]
    assert len(sc1_loads) >= 2, "token loads of the tail are sc1"
