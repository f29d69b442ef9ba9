"""End-to-end parity of the HIP model against (a) the golden fixtures produced by the reference itself and (b) the CPU
oracle on fresh seeded inputs.  Everything goes through cm3p_amd's public model class, i.e. through the C ABI.

Tolerances (stated once): the GPU path uses bf16 GEMM/attention operands with fp32 accumulation, fp32 residual stream,
norm statistics, softmax and head, against an all-fp32 reference:
  - integer outputs (audio slot indices, variation targets): bit-exact;
  - hidden states / embeddings: relative L2 error <= 2e-2;  logits: <= 3e-2 relative L2;
  - gradients: relative L2 error <= 6e-2 per tensor (two bf16 roundings per GEMM input on the way back);
  - loss: |diff| <= 3e-2 on the O(1)-scale golden weights (attention logits are deliberately sharp there) and
    <= 1e-3 at reference-init scale (the north-star bound), checked in test_loss_within_1e3_at_reference_init.
"""
import os

import pytest
import torch
from safetensors.torch import load_file

from cases import CASES

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
GPU_CASES = [n for n in CASES if n.startswith("d64") or n.startswith("c1")]  # c1: head_dim 16, on the generic attention kernels since r04


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


# Fixture tolerances.  FIX_TOL is the class bound of r01-r04 (the O(1)-scale golden weights make attention logits deliberately sharp: the
# worst case of a class sits at a third to a half of it).  Since r05 every single (case, quantity) is ALSO held to 3 x the error measured for
# it on MI355X (tests/golden/fixture_errors_r05.json, written by this very test; the kernels are bit-reproducible, so a run repeats its
# errors exactly): the typical case measures 3-10 x below its class bound, and a 5 x regression there used to pass.
FIX_TOL = dict(loss=3e-2, logits=3e-2, embeds=2e-2, pooled=2e-2, hidden=2e-2, mlm_logits=3e-2, audio=2e-2, grad=6e-2)
FIX_MEASURED: dict = {}
try:
    import json as _json

    _fix_file = _json.load(open(os.path.join(GOLD, "fixture_errors_r05.json")))
    FIX_BASE = _fix_file["measured"]
    FIX_BASE_TOOLCHAIN = _fix_file.get("toolchain")  # (absent in the r05 file: it was measured with the toolchain named below)
except OSError:
    FIX_BASE, FIX_BASE_TOOLCHAIN = {}, None
FIX_R05_TOOLCHAIN = {"hip": "7.0", "torch": "2.10.0+rocm7.0"}  # what tests/golden/fixture_errors_r05.json was measured with


def _toolchain() -> dict:
    return {"hip": ".".join(str(torch.version.hip or "").split(".")[:2]), "torch": torch.__version__}


def _fix(tag: str, key: str, value: float, bound: str):
    """records the error and asserts it against min(class bound, 3 x the committed measurement of this case and quantity).  The tight
    per-case bound is a SELF-measurement (one MI355X, one hipcc build; r05 advisor): it is floored at a tenth of the class bound -
    quantities measured near 1e-6 would otherwise get an absolute bound a different FMA contraction can exceed with nothing wrong - and
    applies only under the toolchain the file was measured with; any other toolchain is held to the class bound alone."""
    FIX_MEASURED.setdefault(tag, {})[key] = float(value)
    tol = FIX_TOL[bound]
    base = FIX_BASE.get(tag, {}).get(key)
    if base is not None and _toolchain() == (FIX_BASE_TOOLCHAIN or FIX_R05_TOOLCHAIN):
        tol = min(tol, max(3.0 * base, 0.1 * FIX_TOL[bound]))
    assert value <= tol, f"{tag}: {key} = {value:.3e} > {tol:.2e}"


@pytest.fixture(scope="module", autouse=True)
def _dump_fixture_errors():
    yield
    import json

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fixture_errors.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        worst: dict = {}
        for tag, d in FIX_MEASURED.items():
            for k, v in d.items():
                b = "grad" if k.startswith("grad.") else k
                worst[b] = max(worst.get(b, 0.0), v)
        with open(path, "w") as f:
            json.dump(dict(tolerances=FIX_TOL, worst=worst, measured=FIX_MEASURED, toolchain=_toolchain()), f, indent=1, sort_keys=True)
    except OSError:
        pass


def _build(name, dtype=torch.float32):
    from cm3p_amd import CM3PConfig, CM3PModel

    cfg = CM3PConfig(**CASES[name]["cfg"])
    model = CM3PModel(cfg)
    sd = load_file(os.path.join(GOLD, "weights_c1.safetensors" if name.startswith("c1") else "weights_d64.safetensors"))
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    sd.update({k[2:]: v for k, v in blob.items() if k.startswith("w.")})  # parameters only this case has (MLM head)
    model.load_state_dict(sd, strict=True)
    return model.to(DEV).to(dtype).train()


def _inputs(blob):
    return {k[3:]: v.to("cuda") for k, v in blob.items() if k.startswith("in.")}  # the CURRENT device (multi-GPU tests set it per rank)


@pytest.mark.parametrize("name", GPU_CASES)
def test_forward_backward_matches_reference_fixture(name):
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name)
    out = model(**_inputs(blob))
    tag = f"{name} padded"
    _fix(tag, "loss", abs(out.loss.item() - blob["loss"].item()), "loss")
    _fix(tag, "logits", _rel(out.logits_per_metadata, blob["logits_per_metadata"]), "logits")
    _fix(tag, "metadata_embeds", _rel(out.metadata_embeds, blob["metadata_embeds"]), "embeds")
    _fix(tag, "beatmap_embeds", _rel(out.beatmap_embeds, blob["beatmap_embeds"]), "embeds")
    _fix(tag, "beatmap_pooled", _rel(out.beatmap_model_output.pooler_output, blob["beatmap_pooler_output"]), "pooled")
    _fix(tag, "metadata_pooled", _rel(out.metadata_model_output.pooler_output, blob["metadata_pooler_output"]), "pooled")
    if "beatmap_last_hidden_state" in blob:
        mask = blob["in.attention_mask"].bool()
        got = out.beatmap_model_output.last_hidden_state.float().cpu()
        assert torch.isfinite(got).all()  # includes padded rows whose local-attention window is empty
        _fix(tag, "hidden", _rel(got[mask], blob["beatmap_last_hidden_state"][mask]), "hidden")
    if "logits" in blob:  # MLM head: (B, S, vocab) logits, compared on the labelled positions and overall
        assert out.logits.shape == blob["logits"].shape
        _fix(tag, "mlm_logits", _rel(out.logits, blob["logits"]), "mlm_logits")
    if "audio_embeds" in blob:
        got_audio = out.beatmap_model_output.audio_model_output.audio_embeds
        assert got_audio.shape == blob["audio_embeds"].shape
        _fix(tag, "audio", _rel(got_audio, blob["audio_embeds"]), "audio")
    # output container: field order and shapes (Trainer consumes it positionally)
    assert list(out.keys())[:5] == ["loss", "logits_per_beatmap", "logits_per_metadata", "metadata_embeds", "beatmap_embeds"]
    lpm = out.logits_per_metadata
    want_lpb = lpm.permute(2, 0, 1) if lpm.dim() == 3 else lpm.t()
    assert torch.equal(out.logits_per_beatmap, want_lpb)

    out.loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for k, v in blob.items():
        if not k.startswith("grad."):
            continue
        g = params[k[5:]].grad
        assert g is not None, k
        if v.norm() < 1e-8:
            assert g.float().norm().item() < 1e-5, k
        else:
            _fix(tag, k, _rel(g, v), "grad")
        checked += 1
    assert checked >= 10
    # nn.Embedding(padding_idx=0): the padding row never receives gradient
    assert params["beatmap_model.encoder.embeddings.tok_embeddings.weight"].grad[0].abs().max().item() == 0.0


def test_loss_within_1e3_at_reference_init():
    """North-star bound: loss within 1e-3 of the reference path.  Oracle (CPU fp32) vs HIP on the same seeded
    reference-init weights and synthetic batch, default-architecture towers shortened to 4 / 2 layers to keep the CPU
    side in seconds."""
    from cm3p_amd import CM3PConfig, CM3PModel
    from oracle import cm3p_oracle as O

    cfg = dict(
        beatmap_config=dict(num_hidden_layers=4, cls_embed=False, audio_config=dict(num_hidden_layers=1)),
        metadata_config=dict(num_hidden_layers=2, cls_embed=False),
    )
    sd = O.init_state_dict(cfg, seed=0)
    batch = O.synthetic_batch(cfg, B=8, S=512, L=64, seed=1234, padded=True)
    with torch.no_grad():
        want = O.forward(sd, cfg, **batch)
    model = CM3PModel(CM3PConfig(**cfg))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV)
    with torch.no_grad():
        out = model(**{k: v.to(DEV) for k, v in batch.items()})
    assert abs(out.loss.item() - want["loss"].item()) <= 1e-3, (out.loss.item(), want["loss"].item())
    assert (out.logits_per_metadata.cpu() - want["logits_per_metadata"]).abs().max().item() <= 2e-2


def test_bf16_model_and_eval_mode():
    """`model.to(bf16)` (how the reference's tests run on GPU, ref:tests/test_cm3p.py:12-14) and no-grad inference."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name, torch.bfloat16).eval()
    with torch.no_grad():
        out = model(**_inputs(blob))
    assert abs(out.loss.item() - blob["loss"].item()) <= 6e-2
    assert _rel(out.logits_per_metadata, blob["logits_per_metadata"]) <= 6e-2
    model.train()
    out = model(**_inputs(blob))
    out.loss.backward()
    g = model.beatmap_model.encoder.layers[1].attn.Wqkv.weight.grad
    assert g.dtype == torch.bfloat16 and torch.isfinite(g.float()).all()


def test_metadata_tower_on_its_own_stream_gives_the_same_step(monkeypatch):
    """On one rank the metadata tower runs on a second stream beside the beatmap tower (CM3PModel._overlap_towers); the same
    kernels on the same data in another stream order: loss and every gradient bit-identical to the single-stream run, also over
    several steps that free and reuse the second stream's memory."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CM3P_TOWER_OVERLAP", mode)
        model = _build(name)
        assert model._overlap_towers(_inputs(blob)["input_ids"], _inputs(blob)["metadata_ids"]) == (mode == "1")
        for _ in range(3):
            for p_ in model.parameters():
                p_.grad = None
            out = model(**_inputs(blob))
            out.loss.backward()
        torch.cuda.synchronize()
        res[mode] = (out.loss.detach().clone(), {k: p_.grad.clone() for k, p_ in model.named_parameters() if p_.grad is not None},
                     out.metadata_embeds.detach().clone())
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][2], res["1"][2])
    assert res["0"][1].keys() == res["1"][1].keys()
    for k in res["0"][1]:  # every parameter, the embedding tables included (their backward visits the tokens in id order: no atomics)
        assert torch.equal(res["0"][1][k], res["1"][1][k]), k


def test_forward_only_calls_fuse_geglu_into_the_wi_gemm_without_changing_a_bit(monkeypatch):
    """Default config, 8 x 4096 beatmap tokens under no_grad: the beatmap tower's Wi GEMMs take the fused kernel
    (encoder._EncoderLayerFn.forward, kernels.gemm_geglu); CM3P_GEGLU_FUSED=0 keeps GEMM + geglu_fwd.  Same embeddings to the bit;
    a training step is unaffected (it needs h and g)."""
    from cm3p_amd import CM3PConfig, CM3PModel, _lib
    from cm3p_amd.synthetic import synthetic_batch

    cfg = CM3PConfig(beatmap_config=dict(cls_embed=False, num_hidden_layers=4), metadata_config=dict(cls_embed=False, num_hidden_layers=2))
    torch.manual_seed(0)
    model = CM3PModel(cfg).to(DEV).eval()
    b = {k: v.to(DEV) for k, v in synthetic_batch(cfg, 8, 4096, 256, seed=3).items()}
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CM3P_GEGLU_FUSED", mode)
        with torch.no_grad():
            _lib.profile_begin()
            o = model(input_ids=b["input_ids"], attention_mask=b.get("attention_mask"), return_loss=False)
            tags = _lib.profile_end()
        outs[mode] = o.beatmap_embeds.clone()
        assert any("gemm8p_kernel<true, true, 6" in t for t in tags) == (mode == "1"), sorted(tags)
        assert any(t.startswith("cm3p_geglu_fwd") for t in tags) == (mode == "0"), sorted(tags)
    assert torch.equal(outs["0"], outs["1"])


def test_forward_only_calls_reuse_bf16_weights_until_the_weight_changes():
    """No-grad calls keep the bf16 copies of the master weights (encoder._bf16_weight_cached) instead of re-casting all of them
    per call; an in-place update (what an optimizer step or load_state_dict does) or a swapped `.data` must be seen at once."""
    from cm3p_amd import encoder as E

    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name).eval()
    E._eval_weights.clear()
    with torch.no_grad():
        a = model(**_inputs(blob)).logits_per_metadata.clone()
        n_cached = len(E._eval_weights)
        b = model(**_inputs(blob)).logits_per_metadata.clone()
        assert n_cached > 0 and len(E._eval_weights) == n_cached and torch.equal(a, b)  # second call: every weight was a hit
        w = model.beatmap_model.encoder.layers[1].attn.Wqkv.weight
        w.mul_(1.5)  # version counter moves
        c = model(**_inputs(blob)).logits_per_metadata.clone()
        E._eval_weights.clear()
        c_fresh = model(**_inputs(blob)).logits_per_metadata.clone()
        assert torch.equal(c, c_fresh) and not torch.equal(c, a)
        w2 = model.metadata_model.encoder.layers[0].mlp.Wi.weight
        w2.data = w2.data * 0.5  # new storage, same version counter
        d = model(**_inputs(blob)).logits_per_metadata.clone()
        E._eval_weights.clear()
        d_fresh = model(**_inputs(blob)).logits_per_metadata.clone()
        assert torch.equal(d, d_fresh) and not torch.equal(d, c)
    # a training call does not read the cache (its casts also make the transposed copies) and is unaffected by it
    model.train()
    out = model(**_inputs(blob))
    out.loss.backward()
    assert torch.isfinite(out.loss)


def test_evaluation_after_an_optimizer_step_sees_the_new_weights():
    """eval -> training step with cm3p_amd.Muon (which writes the weights through raw addresses: no torch op touches them) -> eval:
    the second evaluation must run on the updated weights, i.e. equal a run whose cache was emptied by hand.  The same for an
    optimizer step with no training forward of its own in between (gradients left over from an earlier backward) and for a torch
    optimizer."""
    from cm3p_amd import encoder as E
    from cm3p_amd.muon import Muon

    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    adamw = [p for n, p in named if "embed" in n.lower() or p.ndim <= 1]
    ids = {id(p) for p in adamw}
    opt = Muon(muon_params=[p for _, p in named if id(p) not in ids], lr=0.05, adamw_params=adamw, adamw_lr=1e-2)

    def evaluate():
        model.eval()
        with torch.no_grad():
            return model(**_inputs(blob)).logits_per_metadata.clone()

    def evaluate_uncached():
        E.invalidate_weight_cache()
        return evaluate()

    first = evaluate()
    assert len(E._eval_weights) > 0
    model.train()
    model(**_inputs(blob)).loss.backward()
    opt.step()
    second = evaluate()
    assert torch.equal(second, evaluate_uncached()) and not torch.equal(second, first)
    # a step between two evaluations with NO training forward in between (the gradients are still there): only the version
    # counters that Muon.step bumps say that the weights moved
    assert len(E._eval_weights) > 0
    v0 = model.beatmap_model.encoder.layers[1].attn.Wqkv.weight._version
    opt.step()
    assert model.beatmap_model.encoder.layers[1].attn.Wqkv.weight._version > v0
    third = evaluate()
    assert torch.equal(third, evaluate_uncached()) and not torch.equal(third, second)
    # a torch optimizer
    sgd = torch.optim.SGD(model.parameters(), lr=0.5)
    sgd.step()
    fourth = evaluate()
    assert torch.equal(fourth, evaluate_uncached()) and not torch.equal(fourth, third)
    # a write through `.data` leaves no trace on the parameter: the documented way is invalidate_weight_cache(), and a mode flip
    # (what HF Trainer does around every evaluation) empties the cache as well
    w = model.beatmap_model.encoder.layers[0].mlp.Wo.weight
    w.data.mul_(1.25)
    model.train()
    fifth = evaluate()
    assert torch.equal(fifth, evaluate_uncached()) and not torch.equal(fifth, fourth)


def test_inputs_on_cpu_are_refused():
    from cm3p_amd import CM3PConfig, CM3PModel

    model = CM3PModel(CM3PConfig(**CASES["d64_cls_nopad"]["cfg"])).to(DEV)
    ids = torch.ones(2, 16, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        model(input_ids=ids, metadata_ids=ids)


def test_trainer_drives_the_model_two_steps(tmp_path):
    """HF Trainer (the reference's train.py harness, ref:train.py:360-375) steps the model: bf16 autocast, default collator."""
    from transformers import Trainer, TrainingArguments

    from cm3p_amd import CM3PConfig, CM3PModel
    from cases import make_inputs

    model = CM3PModel(CM3PConfig(**CASES["d64_mean_pad"]["cfg"]))
    batch = make_inputs("d64_mean_pad")

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            return {k: v[i % v.shape[0]] for k, v in batch.items()}

    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=4, max_steps=2, bf16=True, report_to=[],
                             logging_steps=1, save_strategy="no", learning_rate=1e-4, dataloader_num_workers=0,
                             remove_unused_columns=False)
    w0 = model.beatmap_model.encoder.layers[0].mlp.Wi.weight.detach().clone()
    trainer = Trainer(model=model, args=args, train_dataset=DS())
    result = trainer.train()
    assert result.global_step == 2 and torch.isfinite(torch.tensor(result.training_loss))
    assert not torch.equal(w0.to(model.device), model.beatmap_model.encoder.layers[0].mlp.Wi.weight.detach())


def test_gather_negatives_world_size_one_equals_local_loss():
    """The RCCL gather path with a single rank must reproduce the rank-local loss and gradients exactly."""
    import torch.distributed as dist

    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    try:
        model = _build(name)
        out0 = model(**_inputs(blob))
        out0.loss.backward()
        g0 = model.beatmap_model.encoder.layers[0].attn.Wqkv.weight.grad.clone()
        s0 = model.logit_scale.grad.clone()
        model.zero_grad(set_to_none=True)
        model.gather_negatives = True
        out1 = model(**_inputs(blob))
        out1.loss.backward()
        assert abs(out0.loss.item() - out1.loss.item()) <= 1e-6
        assert torch.allclose(out0.logits_per_metadata, out1.logits_per_metadata, atol=1e-6)
        assert torch.allclose(out0.logits_per_beatmap, out1.logits_per_beatmap, atol=1e-6)
        assert _rel(model.beatmap_model.encoder.layers[0].attn.Wqkv.weight.grad, g0) <= 5e-3  # bf16 re-rounding of grads
        assert abs(model.logit_scale.grad.item() - s0.item()) <= 1e-4
        # DistributedDataParallel over RCCL with the bf16 gradient-compression hook (bench.py --grad-compress bf16 halves the
        # 545 MB fp32 all-reduce): same gradients up to one bf16 rounding
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

        model.zero_grad(set_to_none=True)
        for p in model.beatmap_model.audio_encoder.parameters():
            p.requires_grad_(False)  # no input_features in this case: DDP needs every trainable parameter to get a gradient
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[torch.cuda.current_device()], bucket_cap_mb=1)
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
        out2 = ddp(**_inputs(blob))
        out2.loss.backward()
        torch.cuda.synchronize()
        g2 = model.beatmap_model.encoder.layers[0].attn.Wqkv.weight.grad
        assert torch.isfinite(g2).all() and _rel(g2, g0) <= 1e-2
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, name, out, backend="gloo"):
    import datetime

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0  # gloo: two ranks share the one GPU of the test box; nccl (RCCL): one GPU per rank
    torch.cuda.set_device(dev)
    kw = dict(device_id=torch.device("cuda", dev)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300), **kw)
    try:
        blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
        model = _build(name)
        model.gather_negatives = True
        for p in model.beatmap_model.audio_encoder.parameters():
            p.requires_grad_(False)  # no input_features in this case: DDP needs every trainable parameter to get a gradient
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev])
        full = _inputs(blob)
        b = full["input_ids"].shape[0] // world
        part = {k: v[rank * b:(rank + 1) * b].contiguous() for k, v in full.items()}
        o = ddp(**part)
        o.loss.backward()
        torch.cuda.synchronize()
        out[rank] = (o.loss.item(), model.beatmap_model.encoder.layers[1].attn.Wqkv.weight.grad.cpu(),
                     model.logit_scale.grad.cpu(), model.metadata_projection.weight.grad.cpu(), tuple(o.logits_per_metadata.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_gathered_step_equals_single_process_global_batch(backend):
    """SURVEY §8e on the real kernels: 2 ranks, DDP gradient averaging + all-gathered negatives (the async gather started under
    the metadata tower, its reduce-scatter backward) == one process on the concatenated batch.  gloo: both ranks share cuda:0
    (runs on every box); nccl: RCCL with one GPU per rank - skipped unless the box has two GPUs, so the first multi-GPU box that
    runs the suite exercises RCCL at N > 1 before the scaling bench does.
    Tolerance: bf16 re-rounding between the two decompositions (1e-2)."""
    import socket

    import torch.multiprocessing as mp

    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    name = "d64_cls_nopad"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_two_rank_worker, args=(2, port, name, out, backend), nprocs=2, join=True)

    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name)
    o = model(**_inputs(blob))
    o.loss.backward()
    losses = [out[r][0] for r in range(2)]
    assert abs(sum(losses) / 2 - o.loss.item()) <= 2e-3
    for r in range(2):
        assert out[r][4] == (2, 4)  # this rank's 2 metadata rows against all 4 beatmaps
        assert _rel(out[r][1], model.beatmap_model.encoder.layers[1].attn.Wqkv.weight.grad) <= 2e-2
        assert abs(out[r][2].item() - model.logit_scale.grad.item()) <= 2e-3 * max(1.0, abs(model.logit_scale.grad.item()))
        assert _rel(out[r][3], model.metadata_projection.weight.grad) <= 2e-2


# ------------------------------------------------------------------------------------------------- unpadded execution
@pytest.mark.parametrize("name", ["d64_mean_pad", "d64_mean_longpad", "d64_ragged", "d64_mlm", "d64_audio"])
def test_unpadded_execution_matches_the_reference_fixture(name):
    """unpad_inputs=True packs the valid tokens (the reference's flash_attention_2 path, ref:cm3p/modeling_cm3p.py:911-931) and must
    give the reference's results on every VALID position, the same pooled outputs / loss / gradients (same tolerances as the
    padded path), and zeros at the padding positions of last_hidden_state (= _pad_cm3p_output)."""
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    from cm3p_amd import _lib

    model = _build(name)
    model.unpad_inputs = True
    _lib.profile_begin()
    out = model(**_inputs(blob))
    tags = set(_lib.profile_end())
    assert any("varlen" in t for t in tags), tags  # the packed kernels really ran (no silent padded fallback)
    tag = f"{name} unpadded"
    _fix(tag, "loss", abs(out.loss.item() - blob["loss"].item()), "loss")
    _fix(tag, "logits", _rel(out.logits_per_metadata, blob["logits_per_metadata"]), "logits")
    _fix(tag, "beatmap_embeds", _rel(out.beatmap_embeds, blob["beatmap_embeds"]), "embeds")
    _fix(tag, "beatmap_pooled", _rel(out.beatmap_model_output.pooler_output, blob["beatmap_pooler_output"]), "pooled")
    mask = blob["in.attention_mask"].bool()
    assert not mask.all(), "fixture must contain padding for this test to mean anything"
    got = out.beatmap_model_output.last_hidden_state.float().cpu()
    if "beatmap_last_hidden_state" in blob:
        _fix(tag, "hidden", _rel(got[mask], blob["beatmap_last_hidden_state"][mask]), "hidden")
    assert got[~mask].abs().max().item() == 0.0  # padding rows are zero-filled, as _pad_cm3p_output does
    if "logits" in blob:
        _fix(tag, "mlm_logits", _rel(out.logits[mask.to(out.logits.device)], blob["logits"][mask]), "mlm_logits")
    out.loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for k, v in blob.items():
        if not k.startswith("grad."):
            continue
        g = params[k[5:]].grad
        assert g is not None, k
        if v.norm() < 1e-8:
            assert g.float().norm().item() < 1e-5, k
        else:
            _fix(tag, k, _rel(g, v), "grad")
        checked += 1
    assert checked >= 10


def _unpad_like_the_reference(ids, mask):
    """_unpad_cm3p_input (ref:cm3p/modeling_cm3p.py:88-104) restated: -> unpadded ids, indices, cu_seqlens, max_seqlen."""
    lens = mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(mask.flatten(), as_tuple=False).flatten()
    cu = torch.nn.functional.pad(torch.cumsum(lens, dim=0, dtype=torch.int32), (1, 0))
    return ids.flatten()[indices], indices, cu, int(lens.max())


@pytest.mark.parametrize("name", ["d64_audio", "d64_cls_nopad"])
def test_caller_supplied_unpadded_inputs(name):
    """The reference's forward also takes inputs the CALLER has unpadded (input_ids (total_nnz,), indices, cu_seqlens, max_seqlen,
    batch_size, seq_len; ref:cm3p/modeling_cm3p.py:911-931): same loss / embeddings as the padded fixture, last_hidden_state stays
    (total_nnz, H), CLS pooling reads rows cu_seqlens[:-1] (:624-627)."""
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    inp = _inputs(blob)
    ids_u, indices, cu, max_s = _unpad_like_the_reference(inp["input_ids"], inp["attention_mask"])
    B, S = inp["input_ids"].shape
    model = _build(name)
    extra = {k: v for k, v in inp.items() if k not in ("input_ids", "attention_mask")}
    out = model(input_ids=ids_u, indices=indices, cu_seqlens=cu, max_seqlen=max_s, batch_size=B, seq_len=S, **extra)
    assert abs(out.loss.item() - blob["loss"].item()) <= 3e-2, (out.loss.item(), blob["loss"].item())
    assert _rel(out.beatmap_embeds, blob["beatmap_embeds"]) <= 2e-2
    assert _rel(out.logits_per_metadata, blob["logits_per_metadata"]) <= 3e-2
    h = out.beatmap_model_output.last_hidden_state
    assert h.shape == (ids_u.numel(), CASES[name]["cfg"]["beatmap_config"]["hidden_size"])
    if "beatmap_last_hidden_state" in blob:
        want = blob["beatmap_last_hidden_state"][blob["in.attention_mask"].bool()]
        assert _rel(h, want) <= 2e-2
    out.loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for k, v in blob.items():
        if k.startswith("grad.") and v.norm() >= 1e-8:
            assert _rel(params[k[5:]].grad, v) <= 6e-2, k
            checked += 1
    assert checked >= 8
    # malformed descriptions are refused on the host (they would send the kernels past the rows)
    with pytest.raises(ValueError):
        model(input_ids=ids_u, indices=indices, cu_seqlens=cu[:-1], max_seqlen=max_s, batch_size=B, seq_len=S, **extra)
    with pytest.raises(ValueError):
        model(input_ids=ids_u, indices=indices, cu_seqlens=cu, max_seqlen=1, batch_size=B, seq_len=S, **extra)


def test_caller_unpadded_mlm_logits_repad_is_validated_and_keeps_grad_like_the_reference():
    """MLM head on caller-unpadded rows: logits come back re-padded to (B, S, V) (ref:cm3p/modeling_cm3p.py:999-1001) and equal the
    padded run on the valid positions, zeros elsewhere; a wrong `indices` (too few entries, an entry outside batch_size * seq_len)
    is a ValueError, not an out-of-bounds scatter on the GPU; the re-padded logits stay differentiable exactly when the
    reference's do (labels is None or repad_logits_with_grad)."""
    import copy

    from cm3p_amd import CM3PConfig, CM3PModel

    blob = load_file(os.path.join(GOLD, "d64_mlm.safetensors"))
    inp = _inputs(blob)
    cfg = copy.deepcopy(CASES["d64_mlm"]["cfg"])
    cfg["beatmap_config"]["cls_embed"] = True  # (the reference has no pooling of unpadded rows without a CLS token)
    torch.manual_seed(0)
    model = CM3PModel(CM3PConfig(**cfg)).to(DEV)
    ids_u, indices, cu, max_s = _unpad_like_the_reference(inp["input_ids"], inp["attention_mask"])
    B, S = inp["input_ids"].shape
    extra = {k: v for k, v in inp.items() if k not in ("input_ids", "attention_mask", "labels")}
    want = model(**{k: v for k, v in inp.items() if k != "labels"}).logits.detach()
    mask = inp["attention_mask"].bool()
    labels_u = inp["labels"].flatten()[indices]
    out = model(input_ids=ids_u, indices=indices, cu_seqlens=cu, max_seqlen=max_s, batch_size=B, seq_len=S, labels=labels_u, **extra)
    assert out.logits.shape == want.shape and not out.logits.requires_grad  # labels given: the reference re-pads under no_grad
    assert _rel(out.logits[mask], want[mask]) <= 2e-3
    assert out.logits[~mask].abs().max().item() == 0.0  # padding positions are zeros, as _pad_cm3p_output leaves them
    out2 = model(input_ids=ids_u, indices=indices, cu_seqlens=cu, max_seqlen=max_s, batch_size=B, seq_len=S, **extra)
    assert out2.logits.requires_grad  # labels is None: the reference re-pads with grad
    out2.logits.float().square().mean().backward()
    assert model.decoder.weight.grad is not None and torch.isfinite(model.decoder.weight.grad).all() and model.decoder.weight.grad.abs().sum() > 0
    with pytest.raises(ValueError, match="indices"):
        model(input_ids=ids_u, indices=indices[:-1], cu_seqlens=cu, max_seqlen=max_s, batch_size=B, seq_len=S, **extra)
    bad = indices.clone()
    bad[-1] = B * S
    with pytest.raises(ValueError, match="indices"):
        model(input_ids=ids_u, indices=bad, cu_seqlens=cu, max_seqlen=max_s, batch_size=B, seq_len=S, **extra)


def test_caller_supplied_unpadded_inputs_with_mean_pooling_raise_like_the_reference():
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    inp = _inputs(blob)
    ids_u, indices, cu, max_s = _unpad_like_the_reference(inp["input_ids"], inp["attention_mask"])
    model = _build(name)
    with pytest.raises(NotImplementedError, match="Pooling with unpadded input"):
        model(input_ids=ids_u, indices=indices, cu_seqlens=cu, max_seqlen=max_s, metadata_ids=inp["metadata_ids"],
              metadata_attention_mask=inp["metadata_attention_mask"])


def test_unpadded_and_padded_paths_agree_and_full_batches_stay_padded():
    """Same kernels on the same valid tokens: packed and padded execution agree to bf16 noise; a batch without padding is not
    repacked (nothing to gain) and still runs."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    a, b = _build(name), _build(name)
    b.unpad_inputs = True
    with torch.no_grad():
        oa, ob = a(**_inputs(blob)), b(**_inputs(blob))
    assert _rel(ob.beatmap_embeds, oa.beatmap_embeds) <= 5e-3
    assert b.metadata_model.unpad_inputs is True and _rel(ob.metadata_embeds, oa.metadata_embeds) <= 5e-3  # both towers
    assert abs(ob.loss.item() - oa.loss.item()) <= 2e-3
    full = load_file(os.path.join(GOLD, "d64_cls_nopad.safetensors"))
    c = _build("d64_cls_nopad")
    c.unpad_inputs = True
    with torch.no_grad():
        oc = c(**_inputs(full))
    assert abs(oc.loss.item() - full["loss"].item()) <= 3e-2


def test_unpadded_audio_placeholders_inside_the_padding_get_zero_gradient():
    """A right-padded row whose valid length ends inside the audio placeholder run: in unpadded execution those placeholder
    tokens are dropped from the packed rows, so the embedding backward never writes their audio rows - they must come back as
    exact zeros (their true gradient), not as uninitialised memory flowing into the projector / audio-encoder gradients."""
    name = "d64_audio"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    inp = _inputs(blob)
    n_audio = int((inp["input_ids"][1] == CASES[name]["cfg"]["beatmap_config"]["audio_token_id"]).sum())
    cut = 1 + n_audio // 2  # row 1 keeps [AUDIO_BOS] + half of its placeholders; the rest sits in the padding
    inp["attention_mask"] = inp["attention_mask"].clone()
    inp["attention_mask"][1, cut:] = 0
    grads = {}
    for unpad in (False, True):
        model = _build(name)
        model.unpad_inputs = unpad
        # poison the allocator's free blocks so that "uninitialised" is visibly wrong
        junk = torch.full((64, 1024, 256), float("nan"), device=DEV)
        del junk
        out = model(**inp)
        out.loss.backward()
        torch.cuda.synchronize()
        grads[unpad] = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()
                        if p.grad is not None and k.startswith("beatmap_model.audio_encoder.")}
        assert grads[unpad] and all(torch.isfinite(g).all() for g in grads[unpad].values())
    for k, g in grads[True].items():
        ref = grads[False][k]
        if ref.norm() > 1e-8:
            assert _rel(g, ref) <= 2e-2, f"{k}: {_rel(g, ref):.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float64])
def test_audio_features_of_any_float_dtype(dtype):
    """The reference's extraction script passes bf16 mel features (ref:extract_beatmap_embeddings.py:228-230); the channel-major
    im2col kernel reads fp32, so other dtypes are converted first (reading bf16 as fp32 would walk off the buffer)."""
    name = "d64_audio"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name).eval()
    inp = _inputs(blob)
    with torch.no_grad():
        want = model(**inp)
        inp["input_features"] = inp["input_features"].to(dtype)
        got = model(**inp)
    tol = 1e-6 if dtype == torch.float64 else 3e-2  # bf16 / fp16 features are themselves rounded inputs
    assert _rel(got.beatmap_embeds, want.beatmap_embeds) <= tol
    assert _rel(got.beatmap_model_output.audio_model_output.audio_embeds, want.beatmap_model_output.audio_model_output.audio_embeds) <= tol


# ------------------------------------------------------------------------------------------------- inference consumers
def test_extraction_and_variation_eval_paths():
    """the two forward-only consumers of the same kernels.
    (1) ref:extract_beatmap_embeddings.py:217-234 calls model(input_ids, attention_mask, return_loss=False) without metadata under
        no_grad and reads outputs.beatmap_embeds; (2) evaluation scores every beatmap against V metadata variations per row
        (ref:configs/train/default.yaml:147 test_metadata_variations) - a (B, V, L) metadata batch with variation classes."""
    name = "d64_variations"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name).eval()
    inp = _inputs(blob)
    with torch.no_grad():
        full = model(**inp)
        only = model(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], return_loss=False)
    assert only.loss is None and only.logits_per_metadata is None and only.metadata_embeds is None
    assert torch.equal(only.beatmap_embeds, full.beatmap_embeds)
    assert _rel(only.beatmap_embeds, blob["beatmap_embeds"]) <= 2e-2

    # many variations per row: shapes, finiteness, and row 0 of the big batch equals the small batch's scores
    B, V, L = inp["metadata_ids"].shape
    reps = 16
    big_ids = inp["metadata_ids"].repeat(1, reps, 1)
    big_mask = inp["metadata_attention_mask"].repeat(1, reps, 1)
    big_cls = inp["metadata_variation_classes"].repeat(1, reps)
    big_cls[:, V:] = big_cls[:, V:].clamp_min(1) * (big_cls[:, V:] != 0) + (big_cls[:, V:] == 0) * 1  # one original per row only
    with torch.no_grad():
        big = model(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], metadata_ids=big_ids,
                    metadata_attention_mask=big_mask, metadata_variation_classes=big_cls, return_loss=True)
    assert big.logits_per_metadata.shape == (B, V * reps, B)
    assert torch.isfinite(big.logits_per_metadata).all() and torch.isfinite(big.loss)
    assert torch.equal(big.logits_per_metadata[:, :V], full.logits_per_metadata)


# ------------------------------------------------------------------------------------------------- stand-alone classes
def _load_into(model, name="d64_mlm"):
    sd = load_file(os.path.join(GOLD, "weights_d64.safetensors"))
    sd.update({k[2:]: v for k, v in load_file(os.path.join(GOLD, f"{name}.safetensors")).items() if k.startswith("w.")})
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=True)
    return model.to(DEV).train()


def test_standalone_classes_match_reference_fixture():
    """CM3PBeatmapModelWithProjection / CM3PMetadataModelWithProjection / CM3PForMaskedLM against outputs of the reference's own
    classes (tests/golden/make_golden_variants.py): same state-dict keys as the matching parts of CM3PModel, un-normalised
    projections, MLM logits / loss / gradients."""
    from cm3p_amd import CM3PConfig
    from cm3p_amd.modeling_cm3p import CM3PBeatmapModelWithProjection, CM3PForMaskedLM, CM3PMetadataModelWithProjection

    gold = load_file(os.path.join(GOLD, "variants_d64.safetensors"))
    blob = load_file(os.path.join(GOLD, "d64_mlm.safetensors"))
    inp = _inputs(blob)
    cfg = CM3PConfig(**CASES["d64_mlm"]["cfg"])

    bm = _load_into(CM3PBeatmapModelWithProjection(cfg.beatmap_config))
    out = bm(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"])
    assert _rel(out.beatmap_embeds, gold["bproj.beatmap_embeds"]) <= 2e-2
    out.beatmap_embeds.square().sum().backward()
    assert _rel(bm.beatmap_projection.weight.grad, gold["bproj.grad.beatmap_projection.weight"]) <= 6e-2
    assert _rel(bm.beatmap_model.encoder.layers[1].attn.Wqkv.weight.grad, gold["bproj.grad.beatmap_model.encoder.layers.1.attn.Wqkv.weight"]) <= 6e-2

    mm = _load_into(CM3PMetadataModelWithProjection(cfg.metadata_config))
    out = mm(input_ids=inp["metadata_ids"], attention_mask=inp["metadata_attention_mask"])
    assert _rel(out.metadata_embeds, gold["mproj.metadata_embeds"]) <= 2e-2

    ml = _load_into(CM3PForMaskedLM(cfg.beatmap_config))
    assert int(gold["mlm.tied"].item()) == int(ml.decoder.weight.data_ptr() == ml.get_input_embeddings().weight.data_ptr())
    out = ml(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], labels=inp["labels"])
    assert abs(out.loss.item() - gold["mlm.loss"].item()) <= 3e-2
    assert out.logits.shape == gold["mlm.logits"].shape and _rel(out.logits, gold["mlm.logits"]) <= 3e-2
    out.loss.backward()
    assert _rel(ml.head.dense.weight.grad, gold["mlm.grad.head.dense.weight"]) <= 6e-2
    assert _rel(ml.decoder.weight.grad, gold["mlm.grad.decoder.weight"]) <= 6e-2
    assert _rel(ml.beatmap_model.encoder.layers[0].mlp.Wi.weight.grad, gold["mlm.grad.beatmap_model.encoder.layers.0.mlp.Wi.weight"]) <= 6e-2


@pytest.mark.parametrize("tag,num_labels", [("cls_ce", 5), ("cls_mse", 1), ("cls_bce", 4)])
def test_classifier_variant_matches_reference_fixture(tag, num_labels):
    """CM3PForBeatmapClassification: CrossEntropy / MSE / BCE-with-logits as HF infers them from num_labels and the label dtype
    (ref:cm3p/modeling_cm3p.py:1196-1218), against the reference's own class."""
    import copy

    from cm3p_amd import CM3PConfig
    from cm3p_amd.modeling_cm3p import CM3PForBeatmapClassification

    gold = load_file(os.path.join(GOLD, "variants_d64.safetensors"))
    blob = load_file(os.path.join(GOLD, "d64_mlm.safetensors"))
    inp = _inputs(blob)
    bc = copy.deepcopy(CM3PConfig(**CASES["d64_mlm"]["cfg"]).beatmap_config)
    bc.num_labels = num_labels
    bc.problem_type = None
    model = CM3PForBeatmapClassification(bc)
    sd = load_file(os.path.join(GOLD, "weights_d64.safetensors"))
    sd = {k: v for k, v in sd.items() if k in model.state_dict()}
    sd["classifier.weight"] = gold[f"{tag}.w.classifier.weight"]
    sd["classifier.bias"] = gold[f"{tag}.w.classifier.bias"]
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    out = model(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], labels=gold[f"{tag}.labels"].to(DEV))
    assert out.logits.shape == gold[f"{tag}.logits"].shape
    assert _rel(out.logits, gold[f"{tag}.logits"]) <= 3e-2
    assert abs(out.loss.item() - gold[f"{tag}.loss"].item()) <= 3e-2
    out.loss.backward()
    assert _rel(model.classifier.weight.grad, gold[f"{tag}.grad.classifier.weight"]) <= 6e-2
    assert _rel(model.classifier.bias.grad, gold[f"{tag}.grad.classifier.bias"]) <= 6e-2
    assert _rel(model.beatmap_model.encoder.final_norm.weight.grad, gold[f"{tag}.grad.beatmap_model.encoder.final_norm.weight"]) <= 6e-2


def test_sparse_prediction_mlm_matches_reference_fixture():
    """CM3PForMaskedLM with sparse_prediction: logits only for the labelled positions, same loss (ref:cm3p/modeling_cm3p.py:1349-1357)."""
    import copy

    from cm3p_amd import CM3PConfig
    from cm3p_amd.modeling_cm3p import CM3PForMaskedLM

    gold = load_file(os.path.join(GOLD, "variants_d64.safetensors"))
    blob = load_file(os.path.join(GOLD, "d64_mlm.safetensors"))
    inp = _inputs(blob)
    bc = copy.deepcopy(CM3PConfig(**CASES["d64_mlm"]["cfg"]).beatmap_config)
    bc.sparse_prediction = True
    ml = _load_into(CM3PForMaskedLM(bc))
    out = ml(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"], labels=inp["labels"])
    assert out.logits.shape == gold["mlm_sparse.logits"].shape
    assert _rel(out.logits, gold["mlm_sparse.logits"]) <= 3e-2
    assert abs(out.loss.item() - gold["mlm_sparse.loss"].item()) <= 3e-2
    out.loss.backward()
    assert _rel(ml.decoder.weight.grad, gold["mlm_sparse.grad.decoder.weight"]) <= 6e-2
    assert _rel(ml.beatmap_model.encoder.layers[0].mlp.Wi.weight.grad, gold["mlm_sparse.grad.beatmap_model.encoder.layers.0.mlp.Wi.weight"]) <= 6e-2


def test_mlm_logits_used_outside_the_loss_still_get_their_gradient():
    """The MLM head and its loss are one autograd node that writes the logits' gradient in bf16 for the decoder GEMMs; a
    gradient that arrives through the returned `logits` as well (not the Trainer's path) is added to it.  Checked against the
    sum of the two separate backward passes."""
    name = "d64_mlm"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    ins = _inputs(blob)
    w = torch.randn(tuple(blob["logits"].shape), generator=torch.Generator().manual_seed(3)).to(DEV) * 1e-3

    def grads(use_loss, use_logits):
        m = _build(name)
        out = m(**ins)
        obj = 0.0
        if use_loss:
            obj = obj + out.loss
        if use_logits:
            obj = obj + (out.logits * w).sum()
        obj.backward()
        return {k: p.grad for k, p in m.named_parameters() if p.grad is not None}

    both, a, b = grads(True, True), grads(True, False), grads(False, True)
    assert set(both) == set(a)
    for k in both:
        want = a[k] + (b[k] if k in b else 0)
        assert _rel(both[k], want) <= 2e-2, (k, _rel(both[k], want))


def test_input_dtype_and_layout_variants_give_the_reference_case_result():
    """What callers actually pass: bool / float attention masks, int32 token ids (nn.Embedding takes IntTensor), non-contiguous
    id tensors, unpadded execution with a bool mask.  All must equal the int64 case bit for bit - and none may reach a kernel
    with a dtype it does not read (int32 ids read as int64 walk off the embedding table: a GPU fault, found in r01)."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    ins = _inputs(blob)
    model = _build(name)
    ref = model(**ins).loss
    variants = {
        "bool masks": dict(attention_mask=ins["attention_mask"].bool(), metadata_attention_mask=ins["metadata_attention_mask"].bool()),
        "float masks": dict(attention_mask=ins["attention_mask"].float(), metadata_attention_mask=ins["metadata_attention_mask"].float()),
        "int32 ids": dict(input_ids=ins["input_ids"].int(), metadata_ids=ins["metadata_ids"].int()),
        "non-contiguous ids": dict(input_ids=ins["input_ids"].t().contiguous().t()),
    }
    for what, kw in variants.items():
        out = model(**{**ins, **kw})
        out.loss.backward()
        assert torch.equal(out.loss, ref), what
    model.unpad_inputs = True
    out = model(**{**ins, "attention_mask": ins["attention_mask"].bool()})
    assert abs(out.loss.item() - ref.item()) <= 1e-5, "unpadded, bool mask"


def test_out_of_range_token_ids_and_labels_do_not_fault():
    """nn.Embedding / cross_entropy raise a device-side assert for ids or labels outside the vocabulary; a hand-written kernel must
    not turn that caller bug into an out-of-bounds access: such ids read as a zero row and get no gradient, such labels are
    ignored.  Everything stays finite and the rows of well-formed samples are untouched."""
    name = "d64_mlm"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    ins = _inputs(blob)
    model = _build(name)
    ref = model(**ins)
    V = model.config.beatmap_config.vocab_size
    bad = {k: v.clone() for k, v in ins.items()}
    bad["input_ids"][0, 3] = V + 1000          # far outside the table
    bad["input_ids"][0, 5] = -7
    bad["labels"][0, 3] = V + 5
    out = model(**bad)
    out.loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out.loss) and torch.isfinite(out.logits).all()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), k
    # samples 1.. of the batch never saw the bad ids: their embeddings are unchanged
    assert torch.equal(out.beatmap_embeds[1:], ref.beatmap_embeds[1:])


def test_kernel_wrappers_refuse_a_dtype_the_kernel_does_not_read():
    from cm3p_amd import kernels as K

    pos = torch.arange(8, device=DEV, dtype=torch.int32)
    inv = torch.ones(32, device=DEV)
    with pytest.raises(TypeError):
        K.rope_table(pos, inv)
    with pytest.raises(TypeError):
        K.gather_rows(torch.zeros(4, 8, device=DEV), torch.zeros(2, device=DEV, dtype=torch.int32))
    with pytest.raises(TypeError):
        K.inv_valid_count(torch.zeros(8, device=DEV, dtype=torch.int32), -100)


def test_gradient_checkpointing_recomputes_bit_identically():
    """model.gradient_checkpointing_enable() (ref:cm3p/modeling_cm3p.py:257 supports_gradient_checkpointing): the stack keeps one
    tensor per layer and recomputes the rest with the same deterministic kernels - losses and gradients are bit-identical."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    a, b = _build(name), _build(name)
    b.gradient_checkpointing_enable()
    assert b.beatmap_model.encoder.gradient_checkpointing and b.metadata_model.encoder.gradient_checkpointing
    la, lb = a(**_inputs(blob)).loss, b(**_inputs(blob)).loss
    la.backward()
    lb.backward()
    assert torch.equal(la, lb)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in pa:
        if pa[k].grad is None:
            assert pb[k].grad is None, k
        elif "tok_embeddings" in k:  # the embedding gradient is a float atomic scatter-add: order-dependent in the last bits
            torch.testing.assert_close(pa[k].grad, pb[k].grad, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(pa[k].grad, pb[k].grad), k


def test_layer_gradients_are_delivered_while_lower_layers_are_still_in_backward():
    """Each encoder layer is its own autograd node: when the top layer's weight gradient lands on its parameter the bottom
    layer's has not been produced yet - this is what lets DistributedDataParallel start a bucket's all-reduce under the
    rest of the backward pass."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name)
    layers = model.beatmap_model.encoder.layers
    seen = {}

    def top_layer_done(p):
        seen["bottom_pending"] = layers[0].attn.Wqkv.weight.grad is None

    layers[len(layers) - 1].mlp.Wo.weight.register_post_accumulate_grad_hook(top_layer_done)
    order = []

    def note(i):
        def hook(p):
            order.append(i)
        return hook

    for i, layer in enumerate(layers):
        layer.attn.Wqkv.weight.register_post_accumulate_grad_hook(note(i))
    model(**_inputs(blob)).loss.backward()
    assert seen == {"bottom_pending": True}
    assert order == list(reversed(range(len(layers))))


def test_frozen_weights_get_no_gradient_and_leave_the_others_unchanged():
    """ref:train.py:34,317-321 freezes towers / subsets of parameters: a frozen weight's gradient GEMM is skipped, the rest
    of the gradients are bit-identical to the all-trainable run, and the chain stops below the lowest trainable layer."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    a, b = _build(name), _build(name)
    enc = b.beatmap_model.encoder
    frozen = [enc.layers[1].mlp.Wi.weight, enc.layers[2].attn_norm.weight, enc.layers[0].attn.Wqkv.weight]
    for p in frozen:
        p.requires_grad_(False)
    for p in b.metadata_model.parameters():  # a whole tower, embeddings included: its chain never starts
        p.requires_grad_(False)
    la, lb = a(**_inputs(blob)).loss, b(**_inputs(blob)).loss
    la.backward()
    lb.backward()
    assert torch.equal(la, lb)
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in pa:
        if not pb[k].requires_grad or pa[k].grad is None:  # (the audio front end is not on this case's path)
            assert pb[k].grad is None, k
        elif "tok_embeddings" in k:
            torch.testing.assert_close(pa[k].grad, pb[k].grad, rtol=1e-5, atol=1e-6)
        elif k.startswith("beatmap_model") or k == "beatmap_projection.weight":
            assert torch.equal(pa[k].grad, pb[k].grad), k
    # frozen embeddings + frozen bottom layer: the layers above still train, nothing reaches the embedding table
    c = _build(name)
    encc = c.beatmap_model.encoder
    for p in list(encc.embeddings.parameters()) + list(encc.layers[0].parameters()):
        p.requires_grad_(False)
    lc = c(**_inputs(blob)).loss
    lc.backward()
    assert torch.equal(la, lc)
    pc = dict(c.named_parameters())
    for k in pa:
        if k.startswith("beatmap_model.encoder.layers.") and not k.startswith("beatmap_model.encoder.layers.0.") and pa[k].grad is not None:
            assert torch.equal(pa[k].grad, pc[k].grad), k
    assert encc.embeddings.tok_embeddings.weight.grad is None and encc.layers[0].attn.Wo.weight.grad is None


@pytest.mark.parametrize("unpad", [False, True])
def test_output_hidden_states_match_the_reference_per_layer(unpad):
    """output_hidden_states=True: L+1 tensors - the embedding output and every layer's output (TF:...modeling_modernbert.py:457-470) -
    against the per-layer states the reference produced for d64_mean_pad; valid positions only (padded rows differ by design
    between the padded and the unpadded execution)."""
    name = "d64_mean_pad"
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    model = _build(name).eval()
    model.unpad_inputs = unpad
    inp = _inputs(blob)
    with torch.no_grad():
        out = model(**inp, output_hidden_states=True)
    hs = out.beatmap_model_output.hidden_states
    L = CASES[name]["cfg"]["beatmap_config"]["num_hidden_layers"]
    assert isinstance(hs, tuple) and len(hs) == L + 1
    mask = blob["in.attention_mask"].bool()
    assert _rel(hs[0].cpu()[mask], blob["beatmap_hidden_emb"][mask]) <= 2e-2
    for li in range(L):
        assert _rel(hs[li + 1].cpu()[mask], blob[f"beatmap_hidden_{li}"][mask]) <= 2e-2, li
    assert len(out.metadata_model_output.hidden_states) == CASES[name]["cfg"]["metadata_config"]["num_hidden_layers"] + 1
    # asking for the states must not change the result
    with torch.no_grad():
        plain = model(**inp)
    assert torch.equal(plain.beatmap_embeds, out.beatmap_embeds)


def test_default_config_full_length_properties():
    """BASELINE.json configs[1] architecture (ModernBERT-base towers) at the full beatmap length S = 4096 - too big for the CPU
    oracle inside a test, so size-independent properties of the domain:
      * rows of a batch never interact: changing row 1 leaves row 0's embedding bit-identical;
      * right-padding + mask is transparent: a row's embedding does not depend on what the padded positions hold, and padded
        and unpadded execution agree on it;
      * the same batch twice gives the same bits (no atomics on the forward path)."""
    from cm3p_amd import CM3PConfig, CM3PModel
    from cm3p_amd.synthetic import synthetic_batch

    cfg = CM3PConfig(beatmap_config=dict(cls_embed=False), metadata_config=dict(cls_embed=False))
    torch.manual_seed(0)
    model = CM3PModel(cfg).to(DEV).eval()
    B, S, L = 2, 4096, 256
    batch = {k: v.to(DEV) for k, v in synthetic_batch(cfg, B, S, L, seed=77, audio_T=None).items()}

    def embeds(ids, mask, unpad=False):
        model.unpad_inputs = unpad
        with torch.no_grad():
            return model(input_ids=ids, attention_mask=mask, return_loss=False).beatmap_embeds

    ids, mask = batch["input_ids"], batch["attention_mask"]
    e_a = embeds(ids, mask)
    assert torch.equal(e_a, embeds(ids, mask))  # deterministic
    ids_b = ids.clone()
    ids_b[1] = torch.roll(ids[1], 17)
    e_b = embeds(ids_b, mask)
    assert torch.equal(e_b[0], e_a[0]) and not torch.equal(e_b[1], e_a[1])  # rows are independent

    n1 = 3001  # row 1 is valid up to n1, the rest is padding
    mask_c = mask.clone()
    mask_c[1, n1:] = 0
    ids_c = ids.clone()
    ids_c[1, n1:] = 0
    ids_d = ids.clone()
    ids_d[1, n1:] = torch.randint(3, 100, (S - n1,), device=DEV)  # garbage under the mask
    e_c, e_d = embeds(ids_c, mask_c), embeds(ids_d, mask_c)
    assert torch.equal(e_c[0], e_a[0])  # the full row is untouched by its neighbour's padding
    assert torch.equal(e_c[1], e_d[1])  # masked positions are invisible
    e_u = embeds(ids_c, mask_c, unpad=True)
    assert _rel(e_u, e_c) <= 5e-3  # packed vs padded execution (different tile alignment for row 1 only)
    assert torch.equal(e_u[0], e_c[0])  # row 0 starts at packed row 0: same tiles, same bits


def test_output_attentions_matches_the_reference_eager_probabilities():
    """output_attentions=True: every tower returns L tensors (B, nh, S, S) fp32 = softmax(scale q k^T + mask) as the reference's
    eager path computes them (oracle: eager_attention_probs) - global and sliding-window layers, key padding (exact zeros on
    invisible keys), a padded query row whose window holds no valid key (uniform 1 / S, the finite additive mask's result),
    rows summing to one; the other outputs are unchanged by the flag."""
    from oracle import cm3p_oracle as O

    name = "d64_mean_longpad"  # right padding long enough that local-attention queries in the padding see no key at all
    blob = load_file(os.path.join(GOLD, f"{name}.safetensors"))
    inp = _inputs(blob)
    model = _build(name).eval()
    with torch.no_grad():
        ref_out = model(**inp)
        out = model(**inp, output_attentions=True)
    assert torch.equal(out.logits_per_metadata, ref_out.logits_per_metadata)
    cfg = O.resolve_config(CASES[name]["cfg"])
    sd = {k: v.float() for k, v in model.state_dict().items()}
    for tower, prefix, c, ids, mask, got in (
            ("beatmap", "beatmap_model.encoder.", cfg["beatmap_config"], inp["input_ids"], inp["attention_mask"], out.beatmap_model_output.attentions),
            ("metadata", "metadata_model.encoder.", cfg["metadata_config"], inp["metadata_ids"], inp["metadata_attention_mask"], out.metadata_model_output.attentions)):
        want = []
        O.encoder({k: v.cpu() for k, v in sd.items()}, prefix, c, input_ids=ids.cpu(), attention_mask=mask.cpu(), attn_out=want)
        assert len(got) == c["num_hidden_layers"] == len(want)
        B, S = ids.shape
        saw_dead = False
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.shape == (B, c["num_attention_heads"], S, S) and g.dtype == torch.float32 and not g.requires_grad
            g = g.cpu()
            assert (g - w).abs().max().item() <= 1e-2, (tower, i, (g - w).abs().max().item())  # bf16 operands upstream against an fp32 reference (measured 4.3e-3)
            assert (g.sum(-1) - 1).abs().max().item() <= 5e-3
            invisible = w == 0
            assert (g[invisible] == 0).all()  # masked keys: exact zeros
            dead = (w - 1.0 / S).abs().max(-1).values < 1e-9
            if dead.any():
                saw_dead = True
                assert (g[dead] - 1.0 / S).abs().max().item() < 1e-7
        if tower == "beatmap":
            assert saw_dead  # the fixture exists for this edge case
