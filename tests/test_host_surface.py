"""Host-side drop-in surface, checked on CPU: names, signatures, field order, state-dict keys, registrations
(SURVEY.md §8b).  The reference's own state-dict keys come from the golden weights file it produced."""
import inspect
import os

import pytest
import torch
from safetensors.torch import load_file

from cases import CASES

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_keys_equal_the_reference():
    from cm3p_amd import CM3PConfig, CM3PModel

    model = CM3PModel(CM3PConfig(**CASES["d64_cls_nopad"]["cfg"]))
    ref = load_file(os.path.join(GOLD, "weights_d64.safetensors"))
    ours = model.state_dict()
    assert sorted(ours) == sorted(ref)
    for k, v in ref.items():
        assert tuple(ours[k].shape) == tuple(v.shape), k


def test_default_config_parameter_count_matches_reference():
    from cm3p_amd import CM3PConfig, CM3PModel

    model = CM3PModel(CM3PConfig())
    assert sum(p.numel() for p in model.parameters()) == 136_288_513  # measured on the reference, SURVEY.md §6


def test_forward_signature_is_what_trainer_introspects():
    from cm3p_amd import CM3PModel

    sig = inspect.signature(CM3PModel.forward)
    names = list(sig.parameters)
    for n in ("input_ids", "input_features", "metadata_ids", "attention_mask", "metadata_attention_mask", "position_ids",
              "inputs_embeds", "metadata_variation_classes", "labels", "return_loss", "output_logits", "kwargs"):
        assert n in names, n
    assert sig.parameters["return_loss"].default is True
    from transformers.utils.generic import can_return_loss, find_labels

    assert can_return_loss(CM3PModel)
    assert "labels" in find_labels(CM3PModel)


def test_output_field_order():
    from cm3p_amd.modeling_cm3p import CM3POutput

    fields = [f.name for f in CM3POutput.__dataclass_fields__.values()]
    assert fields == ["loss", "logits_per_beatmap", "logits_per_metadata", "metadata_embeds", "beatmap_embeds", "logits",
                      "metadata_model_output", "beatmap_model_output"]


def test_config_roundtrip_and_auto_registration(tmp_path):
    from transformers import AutoConfig, AutoModel

    from cm3p_amd import CM3PConfig, CM3PModel

    cfg = CM3PConfig(**CASES["d64_mean_pad"]["cfg"])
    assert cfg.model_type == "CM3P" and cfg.beatmap_config.model_type == "CM3PBeatmap"
    assert cfg.beatmap_config.audio_config.model_type == "CM3PAudio" and cfg.metadata_config.model_type == "CM3PMetadata"
    model = CM3PModel(cfg)
    model.save_pretrained(tmp_path)
    cfg2 = AutoConfig.from_pretrained(tmp_path)
    assert type(cfg2).__name__ == "CM3PConfig" and cfg2.beatmap_config.hidden_size == 128
    assert cfg2.beatmap_config.audio_config.n_mels == 16 and cfg2.metadata_config.cls_embed is False
    model2 = AutoModel.from_pretrained(tmp_path)
    assert type(model2).__name__ == "CM3PModel"
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k


def test_reference_defaults():
    from cm3p_amd import CM3PConfig

    c = CM3PConfig()
    b, m, a = c.beatmap_config, c.metadata_config, c.beatmap_config.audio_config
    assert (b.hidden_size, b.intermediate_size, b.num_hidden_layers, b.num_attention_heads, b.vocab_size) == (768, 1152, 22, 12, 3167)
    assert (m.hidden_size, m.intermediate_size, m.num_hidden_layers, m.num_attention_heads, m.vocab_size) == (256, 512, 6, 4, 1000)
    assert (a.hidden_size, a.num_hidden_layers, a.projector_intermediate_size, a.projector_dim, a.n_mels) == (512, 6, 2048, 768, 80)
    assert c.projection_dim == 512 and abs(c.logit_scale_init_value - 2.6592) < 1e-9
    assert [b.is_global_layer(i) for i in range(7)] == [True, False, False, True, False, False, True] and b.half_window == 64
    assert all(m.is_global_layer(i) for i in range(6))
    assert b.audio_token_id == 3166 and b.pad_token_id == 0


def test_attn_implementation_string_is_accepted_without_extra_packages():
    """train.py copies `attn_implementation: "flash_attention_2"` onto the config (ref:train.py:275); no flash-attn needed."""
    from cm3p_amd import CM3PConfig, CM3PModel

    cfg = CM3PConfig(**CASES["d64_cls_nopad"]["cfg"])
    cfg._attn_implementation = "flash_attention_2"
    CM3PModel(cfg)


def test_errors_mirror_the_reference():
    from cm3p_amd import CM3PConfig, CM3PModel

    cfg = CM3PConfig(**CASES["d64_cls_nopad"]["cfg"])
    model = CM3PModel(cfg)
    ids = torch.ones(2, 3, 8, dtype=torch.int64)
    with pytest.raises(ValueError):  # ref:cm3p/modeling_cm3p.py:904-905
        model(input_ids=torch.ones(2, 8, dtype=torch.int64), metadata_ids=ids)
    with pytest.raises(ValueError):  # ref:cm3p/modeling_cm3p.py:907-908
        model(input_ids=torch.ones(2, 8, dtype=torch.int64), metadata_ids=ids[:, 0], output_logits=True)
    bad = CM3PConfig(**CASES["d64_cls_nopad"]["cfg"])
    bad.metadata_config = object()
    with pytest.raises(TypeError):  # ref:cm3p/modeling_cm3p.py:735-745
        CM3PModel(bad)


def test_no_cpu_fallback():
    from cm3p_amd import CM3PConfig, CM3PModel

    model = CM3PModel(CM3PConfig(**CASES["d64_cls_nopad"]["cfg"]))
    with pytest.raises(RuntimeError, match="no CPU"):
        model(input_ids=torch.ones(2, 16, dtype=torch.int64), metadata_ids=torch.ones(2, 8, dtype=torch.int64))


def test_unsupported_shapes_fail_loudly():
    from cm3p_amd import CM3PConfig, CM3PModel

    # head_dim 16 (BASELINE configs[0], the reference's tiny test configuration) constructs since r04: the generic attention kernels
    m16 = CM3PModel(CM3PConfig(**CASES["c1_tiny_nopad"]["cfg"]))
    assert set(m16.state_dict()) == set(load_file(os.path.join(GOLD, "weights_c1.safetensors")))
    bad = {**CASES["c1_tiny_nopad"]["cfg"]}
    bad["beatmap_config"] = {**bad["beatmap_config"], "num_attention_heads": 8}  # head_dim 8: no kernel
    with pytest.raises(NotImplementedError, match="head_dim"):
        CM3PModel(CM3PConfig(**bad))
    with pytest.raises(NotImplementedError):
        CM3PModel(CM3PConfig(has_decoder_head=True))  # loss_type None: the reference would silently use a causal-LM loss
    m = CM3PModel(CM3PConfig(**CASES["d64_mlm"]["cfg"]))
    assert {"head.dense.weight", "head.norm.weight", "decoder.weight", "decoder.bias"} <= set(m.state_dict())


def test_standalone_classes_and_the_one_later_row():
    """The stand-alone classes carry the same parameter names as the matching parts of CM3PModel (so one checkpoint serves all)."""
    from cm3p_amd import CM3PConfig
    from cm3p_amd.modeling_cm3p import (CM3PBeatmapModelWithProjection, CM3PForBeatmapClassification, CM3PForMaskedLM,
                                        CM3PMetadataModelWithProjection)

    cfg = CM3PConfig(**CASES["d64_mlm"]["cfg"])
    cfg.beatmap_config.num_labels = 3
    assert {"classifier.weight", "classifier.bias"} <= set(CM3PForBeatmapClassification(cfg.beatmap_config).state_dict())
    full = set(load_file(os.path.join(GOLD, "weights_d64.safetensors"))) | {"head.dense.weight", "head.norm.weight", "decoder.weight", "decoder.bias"}
    for cls, c in ((CM3PBeatmapModelWithProjection, cfg.beatmap_config), (CM3PMetadataModelWithProjection, cfg.metadata_config),
                   (CM3PForMaskedLM, cfg.beatmap_config)):
        keys = set(cls(c).state_dict())
        assert keys and keys <= full, sorted(keys - full)


def test_cm3p_package_shim_resolves_like_train_py_imports():
    import importlib
    import sys

    for k in [k for k in sys.modules if k == "cm3p" or k.startswith("cm3p.")]:
        del sys.modules[k]
    cm3p = importlib.import_module("cm3p")
    assert cm3p.CM3PModel.__module__ == "cm3p_amd.modeling_cm3p" and cm3p.CM3PConfig.__module__ == "cm3p_amd.configuration_cm3p"
    from cm3p.modeling_cm3p import CM3PForBeatmapClassification, CM3PForMaskedLM  # noqa: F401  (ref:train.py:15)


def test_synthetic_batch_matches_oracle_generator():
    """bench.py's GPU leg and its cpu_baseline leg see the same seeded data."""
    from cm3p_amd import CM3PConfig
    from cm3p_amd.synthetic import synthetic_batch
    from oracle import cm3p_oracle as O

    a = synthetic_batch(CM3PConfig(), 3, 64, 16, seed=5, padded=True, audio_T=64)
    b = O.synthetic_batch({}, 3, 64, 16, seed=5, padded=True, audio_T=64)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
