"""CPU-side checks (run under -m "not gpu"): the C-ABI library loads and exports every symbol include/cor_amd.h
declares, the product's parameter tree equals the reference's state_dict key inventory, the factory's error
behaviour, and the host-side top-k merge. No compute call is made (no GPU here)."""
import os
import re

import numpy as np
import pytest
import torch

from tests.golden_util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    return os.path.exists(os.path.join(ROOT, "cor_amd", "csrc", "libcor_amd.so"))


def test_library_exports_every_declared_symbol():
    if not _built():
        import __graft_entry__ as g
        g.build()
    from cor_amd import _native
    lib = _native.load()
    hdr = open(os.path.join(ROOT, "include", "cor_amd.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|long)\s+(cor_\w+)\s*\(", hdr, flags=re.M)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libcor_amd.so does not export {name}"
    assert sorted(_native.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.cor_version() >= 1


def test_product_fails_loudly_without_gpu():
    from cor_amd import ops
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 16), torch.zeros(4, 16))
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    m = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskedPooling").eval()
    with pytest.raises(RuntimeError):
        m(query_image_inputs=torch.zeros(1, 3, 1024, 1024), support_image_inputs=torch.zeros(1, 3, 384, 384),
          change_text_inputs=torch.ones(1, 64, dtype=torch.long), support_mask_inputs=torch.zeros(1, 1, 384, 384))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cor_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_state_dict_keys_match_reference():
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    ref = [str(k) for k in load("state_dict_keys_sam_base")["keys"]]
    m = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskedPooling")
    mine = sorted(k for k in m.state_dict() if ".siglip." not in k)
    assert mine == ref
    g = load("toplevel_MaskAdapterPooling")
    m = build_model_with_query_support_feat("sam_base", "ViT-B-16-SigLIP-384", None, None, "MaskAdapterPooling")
    mine = set(m.state_dict())
    refk = set(str(k) for k in g["keys"])
    assert {k for k in refk if ".siglip." not in k} == {k for k in mine if ".siglip." not in k}
    assert {k for k in refk if ".siglip." in k} <= mine      # stand-in SigLIP (2 blocks, no MAP head) is a subset


@pytest.mark.parametrize("kw,msg", [(dict(sam_model="sam_tiny"), "Invalid SAM model"),
                                    (dict(siglip_model="ViT-L-14"), "Invalid SigLIP model"),
                                    (dict(mask_pooling="AvgPool"), "Invalid mask pooling method")])
def test_factory_error_behaviour(kw, msg):
    from cor_amd.lib.build_model import build_model_with_query_support_feat
    args = dict(sam_model="sam_base", siglip_model="ViT-B-16-SigLIP-384", mask_pooling="MaskedPooling")
    args.update(kw)
    with pytest.raises(ValueError, match=msg):
        build_model_with_query_support_feat(**args)


def test_host_topk_merge_matches_oracle():
    from cor_amd import retrieval
    from oracle import retrieval as oret
    rng = np.random.default_rng(0)
    Q = torch.from_numpy(rng.standard_normal((5, 256), dtype=np.float32))
    G = torch.from_numpy(rng.standard_normal((300, 256), dtype=np.float32))
    G[17] = G[250]                       # a tie across shards
    full_s, full_i = oret.similarity_topk(Q, G, 8)
    parts_s, parts_i = [], []
    for r in range(4):
        lo, hi = retrieval.shard_bounds(300, 4, r)
        s, i = oret.similarity_topk(Q, G[lo:hi], 8)
        parts_s.append(s); parts_i.append(i + lo)
    s, i = retrieval.merge_topk_host(parts_s, parts_i, 8)
    assert torch.equal(i, full_i) and torch.allclose(s, full_s)
    s2, i2 = oret.merge_topk(list(zip(parts_s, parts_i)), 8)
    assert torch.equal(i2, full_i)


def test_ops_refuse_operands_off_the_current_device(monkeypatch):
    """kernels are enqueued with raw pointers on the CURRENT device's stream, so the host guard must refuse (before
    any launch) operands on two devices, or on a device that is not the current one. No GPU here: tensor stand-ins."""
    from cor_amd import ops

    class T:
        def __init__(self, index):
            self.is_cuda, self.device = True, torch.device("cuda", index)

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    assert ops._dev(T(0), None, T(0)) == torch.device("cuda", 0)
    with pytest.raises(RuntimeError, match="different devices"):
        ops._dev(T(0), T(1))
    with pytest.raises(RuntimeError, match="current device"):
        ops._dev(T(1), T(1))
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops._dev(torch.zeros(1))


def test_packed_cache_sees_submodule_and_inplace_updates():
    """the packed-weight cache must notice a submodule load_state_dict and in-place parameter updates."""
    from tests.test_gpu_parity import _build
    from cor_amd import config
    gcfg = dict(config.siglip_cfg("ViT-B-16-SigLIP-384"), depth=1, t_depth=1, vocab=64)
    m = _build(1, (0,), gcfg, "MaskedPooling")
    f0 = m._fingerprint()
    assert m._fingerprint() == f0
    with torch.no_grad():
        m.mask_decoder.iou_token.weight.add_(1.0)                       # in-place (optimizer-step-like) update
    f1 = m._fingerprint()
    assert f1 != f0
    m.image_encoder.load_state_dict(m.image_encoder.state_dict())      # submodule load: copy_ bumps every _version
    assert m._fingerprint() != f1


def test_siglip_cfg_per_tower_gelu():
    from cor_amd import config
    g = config.siglip_cfg("ViT-L-16-SigLIP2-384")
    assert g["v_gelu"] == "tanh" and g["t_gelu"] == "tanh" and "gelu" not in g
    g = config.normalize_siglip_cfg(dict(gelu="erf", t_gelu="tanh"))
    assert g["v_gelu"] == "erf" and g["t_gelu"] == "tanh"
    with pytest.raises(ValueError):
        config.normalize_siglip_cfg(dict(v_gelu="swish"))


def test_lib_alias_resolves_the_reference_imports():
    """`sys.modules["lib"] = cor_amd.lib` lets the reference's own import lines resolve unchanged
    (my_test.py:14 `from lib.build_model import build_model_with_query_support_feat`; lib/build_model.py:4-10 sub-module names)."""
    import importlib, sys
    import cor_amd.lib as L
    saved = {k: v for k, v in sys.modules.items() if k == "lib" or k.startswith("lib.")}
    try:
        sys.modules["lib"] = L
        for sub in ("build_model", "sam_with_sup_branch", "support_branch", "sam_model", "support_model", "sam_model.image_encoder",
                    "sam_model.mask_decoder", "sam_model.transformer", "sam_model.my_prompt_encoder", "sam_model.common",
                    "support_model.mask_adapter", "support_model.cir_feature_fuse", "support_model.siglip_openclip"):
            sys.modules["lib." + sub] = importlib.import_module("cor_amd.lib." + sub)
        ns = {}
        exec("from lib.build_model import build_model_with_query_support_feat\n"
             "from lib.sam_with_sup_branch import CirSegModelWithQuerySupportFeat\n"
             "from lib.sam_model.image_encoder import ImageEncoderViT\n"
             "from lib.support_model.mask_adapter import MaskAdapterPooling, MaskedPooling", ns)
        from cor_amd.lib.build_model import build_model_with_query_support_feat as ours
        assert ns["build_model_with_query_support_feat"] is ours
        with pytest.raises(ValueError):
            ns["build_model_with_query_support_feat"](sam_model="sam_tiny")                   # lib/build_model.py:49
    finally:
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_siglip_tokenizer_with_a_locally_trained_sentencepiece_model(tmp_path):
    """cor_amd.tokenizer.siglip_tokenizer (utils/dataloader.py:296 -> open_clip.get_tokenizer -> SigLipTokenizer): the real c4-en
    vocabulary is not available offline, so the wrapper's own logic is exercised on a tiny sentencepiece model trained here with the
    T5 special-token layout (<pad> 0, </s> 1, <unk> 2): canonicalisation (lower-case, punctuation stripped, whitespace collapsed),
    EOS id 1 appended, padding WITH id 1 (SigLIP sets pad = eos, so the pooled last position is always id 1), fixed length 64,
    truncation to 63 pieces + EOS."""
    import sentencepiece as spm
    from cor_amd import tokenizer
    words = ["make", "the", "cat", "dog", "larger", "smaller", "red", "blue", "please", "number", "left", "right", "remove", "add", "a", "of"]
    rng = np.random.default_rng(0)
    corpus = tmp_path / "corpus.txt"
    with open(corpus, "w") as f:
        for _ in range(2000):
            f.write(" ".join(rng.choice(words, size=int(rng.integers(3, 12)))) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "tiny"), vocab_size=48, model_type="unigram", hard_vocab_limit=False,
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1, minloglevel=2)
    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "tiny.model"))
    tok = tokenizer.siglip_tokenizer(str(tmp_path / "tiny.model"))
    t = tok("Make the CAT   larger, please!")
    assert t.dtype == torch.int64 and t.shape == (64,)
    ids = sp.encode("make the cat larger please")
    n = len(ids)
    assert t[:n].tolist() == ids and int(t[n]) == 1 and bool((t[n:] == 1).all())          # pieces, EOS, pad = 1
    assert torch.equal(t, tok("make the cat larger please"))                                # canonicalisation is what made them equal
    assert sp.decode(t[:n].tolist()) == "make the cat larger please"
    long = tok(" ".join(["cat dog"] * 100))
    assert long.shape == (64,) and int(long[63]) == 1 and 1 not in long[:63].tolist()      # 63 pieces + EOS: the pooled position is EOS
    assert tokenizer.canonicalize_text("  A.b,C!!  d\te ") == "abc d e"
    empty = tok("?!")
    assert int(empty[0]) == 1 and bool((empty == 1).all())


# ---- bench.py's launcher: --gpus N must start N ranks, and must refuse a mismatching WORLD_SIZE

def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_launch_plan_and_command():
    b = _bench()
    assert b.launch_plan(1, {}) == ("run", None)
    assert b.launch_plan(8, {}) == ("spawn", 8)                              # python bench.py --gpus 8: start the 8 ranks itself
    assert b.launch_plan(8, {"WORLD_SIZE": "8"}) == ("run", None)            # the driver's torchrun form
    assert b.launch_plan(1, {"WORLD_SIZE": "1"}) == ("run", None)
    for gpus, env in ((8, {"WORLD_SIZE": "1"}), (1, {"WORLD_SIZE": "8"}), (2, {"WORLD_SIZE": "x"}), (0, {})):
        assert b.launch_plan(gpus, env)[0] == "refuse"
    cmd = b.spawn_command(4, ["--gpus", "4", "--steps", "5"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5"]
    assert b.parse(["--gpus", "2", "--backend", "gloo"]).backend == "gloo" and b.parse([]).multimask == 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    assert not re.search(r"^(import|from)\s+(torch|cor_amd)", head, flags=re.M), "the launching parent must not import torch / cor_amd at module level"


def test_bench_config_presets_express_every_baseline_config():
    """bench.py --config N: every BASELINE.json config as a preset - model, batch, gallery rows and the gallery's storage
    type (independent of --dtype); explicit flags still win; the preset table quotes BASELINE.json's strings verbatim."""
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    for n in (1, 2, 3, 4):
        assert bench.BASELINE_CONFIGS[n] == base[n], n
    a = bench.parse([])
    assert (a.sam, a.batch, a.gallery, a.gallery_dtype, a.config_text) == ("sam_base", 32, 100000, "bf16", None)   # the headline line
    a = bench.parse(["--dtype", "f32"])
    assert a.gallery_dtype == "f32"
    a = bench.parse(["--config", "1"])
    assert (a.sam, a.siglip, a.batch, a.gallery, a.gallery_dtype) == ("sam_base", "ViT-B-16-SigLIP-384", 32, 10000, "bf16")
    a = bench.parse(["--config", "2", "--gpus", "8"])
    assert (a.batch, a.gallery, a.gpus) == (32, 100000, 8)
    a = bench.parse(["--config", "3"])
    assert (a.sam, a.siglip, a.batch) == ("sam_large", "ViT-L-16-SigLIP-384", 64)
    a = bench.parse(["--config", "4", "--gpus", "8"])
    assert (a.sam, a.batch, a.gallery, a.gallery_dtype) == ("sam_large", 64, 1000000, "fp16") and a.config_text == base[4]
    a = bench.parse(["--config", "4", "--gallery", "200000", "--gallery-dtype", "bf16", "--batch", "8"])               # explicit flags win
    assert (a.batch, a.gallery, a.gallery_dtype) == (8, 200000, "bf16")


def test_bench_gpus2_gloo_starts_two_ranks_by_itself_and_refuses_a_mismatch():
    """`python bench.py --gpus 2 --backend gloo --launch-check` with NO WORLD_SIZE: the parent starts two ranks through
    torch.distributed.run, they form a gloo group of 2 and rank 0's JSON line is relayed; a WORLD_SIZE that disagrees with
    --gpus exits 2 before anything runs; failing ranks (RCCL without GPUs here) give a non-zero exit code and no result line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j == {"launch_check": True, "gpus": 2, "world_size": 2, "backend": "gloo", "rank_sum": 1}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4 but --gpus 2" in r.stderr and r.stdout.strip() == ""
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_native_load_imports_torch_before_binding_the_library():
    """Pins the fix of bbd0560: ONE HIP runtime per process - torch (which bundles its own libamdhip64) must be
    imported before libcor_amd.so is bound, or the library resolves the system runtime and the first launch answers hipErrorNoDevice."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes\n"
        "assert 'torch' not in sys.modules\n"
        "seen = {}\n"
        "real = ctypes.CDLL\n"
        "def spy(path, *a, **k):\n"
        "    if str(path).endswith('libcor_amd.so'):\n"
        "        seen['torch_first'] = 'torch' in sys.modules\n"
        "    return real(path, *a, **k)\n"
        "ctypes.CDLL = spy\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from cor_amd import _native\n"
        "_native.load()\n"
        "assert seen == {'torch_first': True}, seen\n"
        "print('ok')\n")
    if not _built():
        import __graft_entry__ as g
        g.build()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
