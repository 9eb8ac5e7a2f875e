"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/pcvae.h
declares, the module surface mirrors the reference, shapes are validated, and nothing falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import pivotcvae_amd as pa
from pivotcvae_amd import _hip
from pivotcvae_amd.models.listcvae import UserListCVAEWithPrior
from tests import philox_ref
from tests.helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pcvae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcvae_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.LIB_PATH), "run `python -m pivotcvae_amd.build` (or __graft_entry__.build())"
    handle = ctypes.CDLL(_hip.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/pcvae.h but not exported"
    assert sorted(_hip.SIGNATURES) == syms  # the ctypes table mirrors the header one to one
    assert _hip.lib().pcvae_abi_version() == _hip.ABI_VERSION == 3


def test_host_side_argument_checks_need_no_gpu():
    """EINVAL paths return before any HIP call, so they are testable here."""
    L = _hip.lib()
    assert L.pcvae_catalog_ws_bytes(0, 10, 16, 1) == 0
    assert L.pcvae_catalog_ws_bytes(81920, 1000000, 128, 1) > 81920 * 128 * 4
    rc = L.pcvae_gather_rows(None, 10, 16, None, 1, 1, None, 16, None)  # n_idx = 1
    assert rc == -1 and b"null" in L.pcvae_last_error()
    rc = L.pcvae_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None)
    assert rc == -1


def _tables(N=50, NU=7, D=16):
    return torch.nn.Embedding(N, D), torch.nn.Embedding(NU, D)


def test_registry_and_constructor_contract():
    assert list(pa.PIVOTCVAE_MODELS) == ["pivotcvae_gt_pi", "pivotcvae_pt_pi", "pivotcvae_spt_pi", "pivotcvae_sgt_pi",
                                         "pivotcvae_gt_spi", "pivotcvae_pt_spi", "pivotcvae_spt_spi",
                                         "pivotcvae_sgt_spi"]
    rules = {k: (c.TRAIN_RULE, c.INFER_RULE) for k, c in pa.PIVOTCVAE_MODELS.items()}
    assert all(k == f"pivotcvae_{t}_{i}" for k, (t, i) in rules.items())
    assert pa.PivotCVAE is pa.PIVOTCVAE_MODELS["pivotcvae_gt_pi"]
    e, u = _tables()
    good = dict(enc=[102, 24, 24], psm=[26, 24, 24, 16], scm=[42, 24, 24, 64], prior=[22, 12, 12])
    m = pa.UserPivotCVAE(e, u, 5, 16, 4, 6, good["enc"], good["psm"], good["scm"], good["prior"], False, "cpu")
    for attr in ("device", "candidateFlag", "slate_size", "feature_size", "latent_size", "condition_size", "noUser",
                 "docEmbed", "userEmbed", "forward", "recommend", "generate", "get_prior", "encode", "decode",
                 "pick_pivot", "reparametrize", "get_condition", "get_recommended_item", "sample_encoding", "log",
                 "loss"):
        assert hasattr(m, attr), attr
    assert m.candidateFlag is False and m.noUser is False
    # struct asserts of the reference (models/pivotcvae.py:58-71)
    for key, bad in (("enc", [101, 24, 24]), ("psm", [26, 24, 24, 15]), ("scm", [42, 24, 24, 63]), ("prior", [21, 12])):
        st = dict(good)
        st[key] = bad
        with pytest.raises(AssertionError):
            pa.UserPivotCVAE(e, u, 5, 16, 4, 6, st["enc"], st["psm"], st["scm"], st["prior"], False, "cpu")
    with pytest.raises(AssertionError):  # no_user changes every expected width
        pa.UserPivotCVAE(e, None, 5, 16, 4, 6, good["enc"], good["psm"], good["scm"], good["prior"], True, "cpu")
    with pytest.raises(NotImplementedError):
        pa.UserPivotCVAE(e, u, 5, 16, 4, 6, good["enc"], good["psm"], good["scm"], good["prior"], False, "cpu",
                         fine_tune=True)


@pytest.mark.parametrize("name", ["pivotcvae_gt_pi_user", "pivotcvae_gt_pi_nouser", "listcvae_user", "listcvae_nouser"])
def test_state_dict_keys_match_reference(name):
    g = load(name)
    m, st = g.meta, g.meta["structs"]
    doc = torch.nn.Embedding.from_pretrained(g.t("raw_doc"))
    usr = torch.nn.Embedding.from_pretrained(g.t("raw_user"))
    if m["model"] == "listcvae":
        model = UserListCVAEWithPrior(doc, None if m["no_user"] else usr, m["S"], m["D"], m["Z"], m["S"] + 1,
                                      st["enc"], st["dec"], st["prior"], m["no_user"], "cpu")
    else:
        model = pa.PIVOTCVAE_MODELS[m["model"]](doc, None if m["no_user"] else usr, m["S"], m["D"], m["Z"], m["S"] + 1,
                                               st["enc"], st["psm"], st["scm"], st["prior"], m["no_user"], "cpu")
    assert list(model.state_dict().keys()) == list(g.sd.keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(g.sd[k].shape), k
    # G1: tables are row-normalised copies, frozen
    torch.testing.assert_close(model.docEmbed.weight, g.t("sd/docEmbed.weight"), rtol=2e-6, atol=2e-7)
    assert not model.docEmbed.weight.requires_grad
    model.load_state_dict(g.sd)  # reference checkpoints load
    trainable = sorted(k for k, p in model.named_parameters() if p.requires_grad)
    assert trainable == sorted(k for k in g.sd if not k.startswith(("docEmbed", "userEmbed")))


def test_no_cpu_fallback():
    e, u = _tables()
    m = pa.UserPivotCVAE(e, u, 5, 16, 4, 6, [102, 24, 24], [26, 24, 24, 16], [42, 24, 24, 64], [22, 12, 12], False, "cpu")
    s = torch.zeros(2, 5, dtype=torch.long)
    r = torch.zeros(2, 5)
    uu = torch.zeros(2, 1, dtype=torch.long)
    for call in (lambda: m.forward(s, r, u=uu), lambda: m.recommend(r, uu), lambda: m.get_prior(r, uu),
                 lambda: m.loss(s, r, uu, 0.001)):
        with pytest.raises(RuntimeError, match="ROCm device only"):
            call()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pivotcvae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "/root/reference" not in text, f


def test_philox_reference_vectors():
    """Philox4x32-10 known-answer tests (Random123 kat_vectors) pin the host restatement of the RNG."""
    out = philox_ref.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = philox_ref.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    out = philox_ref.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in out] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    k = philox_ref.keep_mask(64, 4096, 0.1, 99)
    assert abs(k.mean() - 0.1) < 0.005


def test_flat_adam_views_keep_names_and_storage():
    from pivotcvae_amd.optim import FlatAdam
    e, u = _tables()
    m = pa.UserPivotCVAE(e, u, 5, 16, 4, 6, [102, 24, 24], [26, 24, 24, 16], [42, 24, 24, 64], [22, 12, 12], False, "cpu")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = FlatAdam(m, 1e-3)
    assert list(m.state_dict().keys()) == list(before.keys())
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    n = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert opt.flat.numel() == n == opt.grad.numel()
    p0 = next(p for p in m.parameters() if p.requires_grad)
    assert p0.data_ptr() == opt.flat.data_ptr() and p0.grad.data_ptr() == opt.grad.data_ptr()


def test_lds_bank_model_of_the_catalog_kernels():
    """tools/lds_bank_check.py: the swizzled LDS image is conflict-free for the row reads and the transposed reads of both
    MFMA shapes, the swizzle is an involution and the global_load_lds lane map covers the image exactly once."""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lds_bank_check.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_lds_images_of_the_gemm_kernel():
    """tools/gemm_lds_check.py: the LDS-DMA lane -> source map of the MLP GEMMs covers both operand images exactly once, lands
    every element where the MFMA operand reads look for it, and those reads are bank-conflict free."""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_lds_check.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_gemm_desc_layout_matches_the_header(tmp_path):
    """the ctypes mirror of pcvae_gemm_desc has the size and field offsets gcc gives the struct in include/pcvae.h"""
    import subprocess
    fields = [f for f, _ in _hip.GemmDesc._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pcvae.h"\nint main(void) { printf("%zu", sizeof(pcvae_gemm_desc));\n'
                   + "".join(f'printf(" %zu", offsetof(pcvae_gemm_desc, {f}));\n' for f in fields) + "return 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got[0] == ctypes.sizeof(_hip.GemmDesc)
    assert got[1:] == [getattr(_hip.GemmDesc, f).offset for f in fields]


def test_linear_group_argument_checks_need_no_gpu():
    L = _hip.lib()
    assert L.pcvae_linear_group(None, 1, None, 0, None) == -1
    d = (_hip.GemmDesc * 1)(_hip.GemmDesc(_hip.GEMM_FWD, 0, 8, 4, 8, 4, 8, 4, None, 0, None, 3, 4, 5))   # lda < K
    assert L.pcvae_linear_group(d, 1, None, 0, None) == -1 and b"leading" in L.pcvae_last_error()
    d[0].kind = 9
    assert L.pcvae_linear_group(d, 1, None, 0, None) == -1 and b"kind" in L.pcvae_last_error()
    many = (_hip.GemmDesc * 7)()
    assert L.pcvae_linear_group(many, 7, None, 0, None) == -1
    # scratch of the weight gradients: a fixed 64 KB counter region + the partial tiles of every batch split; none without one
    assert L.pcvae_linear_group_ws_bytes(d, 1) == 0
    w = (_hip.GemmDesc * 1)(_hip.GemmDesc(_hip.GEMM_DW, 0, 8, 256, 8, 1419, 8, 1419, None, 0, None, 8192, 256, 1419))
    need = L.pcvae_linear_group_ws_bytes(w, 1)
    tiles = 4 * 23
    assert need > 65536 and (need - 65536) % ((tiles * 4096 + 4 * 64) * 4) == 0   # splits x (partial tiles + bias partials)
    assert L.pcvae_linear_group(w, 1, 8, need - 1, None) == -1 and b"workspace" in L.pcvae_last_error()
    one = (_hip.GemmDesc * 1)(_hip.GemmDesc(_hip.GEMM_DW, 0, 8, 16, 8, 16, 8, 16, None, 0, None, 64, 16, 16))   # one split: counters only
    assert L.pcvae_linear_group_ws_bytes(one, 1) == 65536


def test_pipelined_kernel_steady_state_loop_has_no_compiler_copies():
    """The software-pipelined catalog kernel issues its MFMAs as inline asm; hipcc must not place register copies
    (v_accvgpr_*, v_mov_*, scratch) inside its steady-state loop, where no wait states protect them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_loop_check", os.path.join(ROOT, "tools", "isa_loop_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels = mod.hot_loops()
    assert len(kernels) >= 12, "pipelined kernels not found in the generated ISA"   # 3 CE + 3 split-bf16 CE + 6 screening instantiations
    assert sum("x3" in name for name in kernels) == 3                              # bf16x3 at D = 128 and D = 256, bf16x6 at D = 128
    assert sum("x3_pipe_kernel<128, 2, 3>" in name for name in kernels) == 1       # (round 4) the bf16x6 instantiation
    # no asm MFMA anywhere in these kernels reads a VGPR that a VALU instruction wrote fewer than two wait states earlier (hipcc
    # does not protect inline-asm MFMAs; round 3 met a stale read behind the loop-entry copies of the bf16x3 kernel)
    fresh, n_mfma = mod.mfma_fresh_operand_reads()
    assert n_mfma > 5000 and not fresh, fresh[:5]
    # ... and no VALU instruction / store takes an MFMA result fewer than five wait states behind the MFMA that wrote it
    early, n2 = mod.mfma_result_early_reads()
    assert n2 == n_mfma and not early, early[:5]
    for name, loops in kernels.items():
        assert loops, f"{name}: no steady-state loop found"
        for loop in loops:
            bad = mod.forbidden_in(loop, no_mov="catalog_ce_" in name)
            assert not bad, (name, bad[:5])
            if "x3" in name:
                # an MFMA reads its operands when it starts, and MFMAs queue up in MFMA-dense stretches: no LDS read / VALU result
                # may land in a register that one of the last four MFMAs in front of it reads (catalog_x3.h: x3_keep)
                clob = mod.queued_operand_clobbers(loop)
                assert not clob, (name, clob[:5])


def test_arrival_counters_wait_for_their_partial_stores():
    """The batch splits of a weight gradient and the KL block partials are handed over through device-scope (sc1) stores + an
    integer arrival counter.  The stores must be ACKNOWLEDGED before the counter is bumped: an explicit s_waitcnt vmcnt(0) has to
    sit between the last sc1 store and the global_atomic_add in the generated ISA (a workgroup-scope release emits none)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_loop_check", os.path.join(ROOT, "tools", "isa_loop_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, seen = mod.arrival_counter_waits()
    assert seen >= 3, "hand-offs not found in the generated ISA"   # weight-gradient GEMM + the two KL kernels
    assert not bad, bad[:5]


def test_catalog_kernel_choice_and_range_alignment(monkeypatch):
    """pcvae_catalog_ce_variant mirrors the launch logic (host only): f32 -> 0; bf16 D = 256 -> pipelined always; D = 128 /
    64 -> pipelined on long ranges only, PCVAE_PIPE_MIN_TILES overrides; unsupported shapes -> -1."""
    L = _hip.lib()
    monkeypatch.delenv("PCVAE_PIPE_MIN_TILES", raising=False)
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 128, _hip.PREC_F32) == 0
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 128, _hip.PREC_BF16) == 2      # config 4 on one GPU: 7 8xx tiles per range
    assert L.pcvae_catalog_ce_variant(10240, 1_000_000, 128, _hip.PREC_BF16) == 2      # one 8-GPU shard: 980 tiles per range
    assert L.pcvae_catalog_ce_variant(40960, 100_000, 64, _hip.PREC_BF16) == 2         # config 3: 391 tiles per range (D = 64: pipelined from 256)
    assert L.pcvae_catalog_ce_variant(40960, 20_000, 64, _hip.PREC_BF16) == 1          # a short catalog at D = 64
    assert L.pcvae_catalog_ce_variant(256, 8192, 128, _hip.PREC_BF16) == 1             # a small catalog
    assert L.pcvae_catalog_ce_variant(163840, 10_000_000, 256, _hip.PREC_BF16) == 2    # config 5
    assert L.pcvae_catalog_ce_variant(64, 1000, 256, _hip.PREC_BF16) == 2
    assert L.pcvae_catalog_ce_variant(64, 1000, 32, _hip.PREC_BF16) == -1
    assert L.pcvae_catalog_ce_variant(0, 1000, 128, _hip.PREC_BF16) == -1
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 128, _hip.PREC_BF16X3) == 3
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 128, _hip.PREC_BF16X6) == 4     # bf16x6: D = 128 only
    assert L.pcvae_catalog_ce_variant(163840, 10_000_000, 256, _hip.PREC_BF16X6) == -1
    monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", "1")
    assert L.pcvae_catalog_ce_variant(256, 8192, 128, _hip.PREC_BF16) == 2
    monkeypatch.setenv("PCVAE_PIPE_MIN_TILES", "1000000000")
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 128, _hip.PREC_BF16) == 1
    assert L.pcvae_catalog_ce_variant(81920, 1_000_000, 256, _hip.PREC_BF16) == 2


def test_sparse_keep_stream_is_bernoulli():
    """the host restatement of the sparse path's kept-set stream (geometric gaps per catalog segment): i.i.d. Bernoulli(p)
    marginals - mean, variance of the row counts, and a flat profile over catalog positions (segment boundaries included)"""
    R, N, p = 300, 20000, 0.01
    k = philox_ref.sparse_keep_mask(R, N, p, seed=5, row_offset=11)
    cnt = k.sum(1).astype(np.int64)
    assert abs(cnt.mean() - N * p) < 4 * (N * p / R) ** 0.5
    assert 0.8 * N * p < cnt.var() < 1.25 * N * p
    col = k.astype(np.int64).sum(0).reshape(100, -1).sum(1)           # 100 position buckets of 200 items: each ~ Poisson(R * 200 * p = 600)
    assert np.abs(col - 600).max() < 5 * 600 ** 0.5
    # neighbouring items are independent: P(both kept) = p^2
    both = int((k[:, :-1] & k[:, 1:]).astype(np.int64).sum())
    assert abs(both - R * (N - 1) * p * p) < 5 * (R * (N - 1) * p * p) ** 0.5
    # row_offset is the global row index: a shard equals the corresponding rows of the whole batch
    assert np.array_equal(philox_ref.sparse_keep_mask(100, N, p, seed=5, row_offset=211), k[200:])


def test_flat_adam_places_the_heads_back_to_back():
    """FlatAdam honours model.flat_param_groups(): the two heads of the encoder / of the prior sit back to back in the flat buffer
    (weights with weights, biases with biases, and the same for the gradient views), so that [W_mu ; W_logvar] is ONE strided
    operand; names, shapes, values and state_dict keys are untouched; parameters without a group keep module order."""
    from pivotcvae_amd.optim import FlatAdam
    g = load("pivotcvae_gt_pi_user")
    m, st = g.meta, g.meta["structs"]
    model = pa.PIVOTCVAE_MODELS["pivotcvae_gt_pi"](torch.nn.Embedding.from_pretrained(g.t("raw_doc")),
                                                  torch.nn.Embedding.from_pretrained(g.t("raw_user")), m["S"], m["D"], m["Z"], m["S"] + 1,
                                                  st["enc"], st["psm"], st["scm"], st["prior"], False, "cpu")
    model.load_state_dict(g.sd)
    opt = FlatAdam(model, 1e-3)

    def back_to_back(a, b):
        return b.data_ptr() == a.data_ptr() + a.numel() * 4

    for a, b in ((model.encmu, model.enclogvar), (model.priorMu, model.priorLogvar)):
        assert back_to_back(a.weight.data, b.weight.data) and back_to_back(a.bias.data, b.bias.data)
        assert back_to_back(a.weight.grad, b.weight.grad) and back_to_back(a.bias.grad, b.bias.grad)
        cat = torch.as_strided(a.weight.data, (2 * a.weight.shape[0], a.weight.shape[1]), a.weight.stride())
        assert torch.equal(cat, torch.cat([a.weight.data, b.weight.data]))
    assert sum(p.numel() for p in opt.params) == opt.flat.numel() and len({id(p) for p in opt.params}) == len(opt.params)
    assert [id(p) for p in opt.params][:2] == [id(model.enc_1.weight), id(model.enc_1.bias)]
    for k, v in model.state_dict().items():
        assert torch.equal(v, g.sd[k]), k
    # the PSM never receives a gradient: with weight decay on, its range is stepped with decay 0 (torch.optim.Adam skips grad None)
    m2 = _fresh_model(g)
    opt2 = FlatAdam(m2, 1e-3, weight_decay=0.1)
    assert {wd for _, _, wd in opt2.segments} == {0.0, 0.1}
    assert sum(n for _, n, wd in opt2.segments if wd == 0.0) == sum(p.numel() for p in m2.params_without_grad())
    assert sum(n for _, n, _ in opt2.segments) == opt2.flat.numel()


def _fresh_model(g):
    m, st = g.meta, g.meta["structs"]
    return pa.PIVOTCVAE_MODELS["pivotcvae_gt_pi"](torch.nn.Embedding.from_pretrained(g.t("raw_doc")),
                                                 torch.nn.Embedding.from_pretrained(g.t("raw_user")), m["S"], m["D"], m["Z"], m["S"] + 1,
                                                 st["enc"], st["psm"], st["scm"], st["prior"], False, "cpu")


def test_round6_host_switches_need_no_gpu():
    """round 6's host-side contracts on a CPU-constructed model: the MLP arithmetic switch (and its round-3 alias, also on a module
    pickled before round 6), the explicit gather-rows switch, the trainer's validation of per-step candidate sets and of the draw's id
    range, capture eligibility decided from the current mode"""
    from pivotcvae_amd import ops
    from pivotcvae_amd.train_generative import Trainer
    e, u = _tables()
    st = dict(enc=[102, 24, 24], psm=[26, 24, 24, 16], scm=[42, 24, 24, 64], prior=[22, 12, 12])
    m = pa.UserPivotCVAE(e, u, 5, 16, 4, 6, st["enc"], st["psm"], st["scm"], st["prior"], False, "cpu")
    assert m.mlp_precision == "f32" and m.mlp_x3 is False and m.gather_rows_bf16 is False
    assert ops.MLP_PRECISIONS == ("f32", "bf16x3", "bf16x6") and set(ops.MLP_ARITHMETIC) == set(ops.MLP_PRECISIONS)
    for name in ops.MLP_PRECISIONS:
        assert m.set_mlp_precision(name) is m and m.mlp_precision == name and m.mlp_x3 == (name == "bf16x3")
    assert m.set_mlp_precision("fp32").mlp_precision == "f32"
    with pytest.raises(ValueError):
        m.set_mlp_precision("fp8")
    old = dict(m.__dict__)                      # a module pickled by round <= 5: the boolean, no mlp_precision
    del m.__dict__["mlp_precision"]
    m.__dict__["mlp_x3"] = True
    assert m.mlp_x3 is True
    m.__dict__.clear()
    m.__dict__.update(old)
    assert m.set_gather_rows("bf16").gather_rows_bf16 is True and m.set_gather_rows("f32").gather_rows_bf16 is False
    assert not ops.gather_rows_are_bf16(m.set_gather_rows("bf16"))     # D = 16 has no bf16 table: nothing to select
    with pytest.raises(ValueError):
        m.set_gather_rows("fp8")
    assert [ops.default_mlp_precision(d) for d in ("f32", "bf16", "bf16x3", "bf16x6")] == ["f32", "bf16x3", "bf16x3", "bf16x6"]
    with ops.mlp_arith("bf16x6"):
        assert ops.GemmGroup().mode == 2 and ops.GemmGroup().x3
        with ops.mlp_arith(True):
            assert ops.GemmGroup().mode == 1
        with ops.mlp_arith(False):
            assert ops.GemmGroup().mode == 0 and not ops.GemmGroup().x3
    assert ops.GemmGroup().mode == 0
    # the trainer (no step is taken: no kernel is launched)
    N = e.weight.shape[0]
    tr = Trainer(m, lr=1e-3, beta=0.001, n_candidate=7, n_items=N - 3, capture_graph=True)
    assert tr.n_items == N - 3 and tr._capturable() and tr._mode()[2] == N - 3
    for bad in (0, N + 1):
        with pytest.raises(ValueError):
            Trainer(m, lr=1e-3, beta=0.001, n_candidate=7, n_items=bad)
    with pytest.raises(ValueError):
        Trainer(m, lr=1e-3, beta=0.001, n_candidate=7, n_neg=5)
    s = torch.zeros(6, 5, dtype=torch.long)
    cand, tgt = torch.zeros(6, 5, 7, dtype=torch.long), torch.zeros(6, 5, dtype=torch.long)
    assert Trainer._check_given_sets((cand, tgt), s)[0] is cand
    for wrong in ((cand[:5], tgt[:5]), (cand[:, :4], tgt), (cand, tgt[:, :4]), (cand.reshape(30, 7), tgt), (cand,), cand):
        with pytest.raises(ValueError):
            Trainer._check_given_sets(wrong, s)
    tr.n_candidate = (cand, tgt)
    assert not tr._capturable() and tr._mode()[1] == "given"
    tr2 = Trainer(m, lr=1e-3, beta=0.001, n_neg=N // 2, capture_graph=True)     # keep probability above the sparse kernel's range
    assert tr2.capture_graph and not tr2._capturable()
