"""Configuration classes of the MI355X CM3P build.

Same class names, `model_type` strings, field names, defaults and nesting as the reference
(ref:cm3p/configuration_cm3p.py:10-341) so that reference checkpoints' `config.json`, `train.py`'s
`CM3PConfig(**args.model)` and the Auto* registry keep working.  Defaults live in tables instead of long
signatures; the layer-type / window rules the encoder needs are derived from the reference's 4.55-era fields.
"""
from __future__ import annotations

from transformers import AutoConfig
from transformers.configuration_utils import PretrainedConfig

# encoder fields shared by the three ModernBERT-shaped towers (ref:cm3p/configuration_cm3p.py:26-46,103-120,200-225)
_ENCODER_COMMON = dict(
    hidden_activation="gelu",
    initializer_range=0.02,
    initializer_cutoff_factor=2.0,
    norm_eps=1e-5,
    norm_bias=False,
    attention_bias=False,
    attention_dropout=0.0,
    local_attention=128,
    local_rope_theta=10000.0,
    embedding_dropout=0.0,
    mlp_bias=False,
    mlp_dropout=0.0,
    decoder_bias=True,
    deterministic_flash_attn=False,
    reference_compile=None,
)
_METADATA_FIELDS = dict(
    _ENCODER_COMMON,
    cls_embed=True, projection_dim=512, initializer_factor=1.0,
    vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=6, num_attention_heads=4,
    max_position_embeddings=128, global_rope_theta=10000.0, global_attn_every_n_layers=1,
)
_AUDIO_FIELDS = dict(
    _ENCODER_COMMON,
    hidden_size=512, intermediate_size=1024, num_hidden_layers=6, num_attention_heads=8,
    max_position_embeddings=4096, global_rope_theta=160000.0, global_attn_every_n_layers=3,
    projector_intermediate_size=2048, projector_dim=768, projector_hidden_act="gelu",
    sample_rate=16000, n_ftt=2048, n_mels=80, hop_length=128, f_min=0, f_max=8000, pad_mode="constant",
)
_BEATMAP_FIELDS = dict(
    _ENCODER_COMMON,
    audio_sos_token_id=3164, audio_eos_token_id=3165, audio_token_id=3166,
    cls_embed=True, projection_dim=512, initializer_factor=1.0,
    vocab_size=3167, hidden_size=768, intermediate_size=1152, num_hidden_layers=22, num_attention_heads=12,
    max_position_embeddings=8192, global_rope_theta=160000.0, global_attn_every_n_layers=3,
    classifier_bias=False, classifier_activation="gelu", sparse_prediction=False, sparse_pred_ignore_index=-100,
    repad_logits_with_grad=False,
)
_TOKEN_IDS = dict(pad_token_id=0, bos_token_id=1, eos_token_id=2)


def _split_kwargs(fields: dict, token_ids: dict, kwargs: dict):
    """Pop this tower's known fields (with defaults) out of kwargs; what remains goes to PretrainedConfig."""
    values = {k: kwargs.pop(k, d) for k, d in fields.items()}
    tokens = {k: kwargs.pop(k, d) for k, d in token_ids.items()}
    # 5.x-style derived names that a config.json written by another stack may carry are not stored
    for derived in ("layer_types", "rope_parameters", "sliding_window"):
        kwargs.pop(derived, None)
    return values, tokens


class _EncoderRules:
    """Derived quantities (the rule the installed ModernBertConfig applies,
    TF:models/modernbert/configuration_modernbert.py:113-162).  Plain mixin: no fields, no annotations."""

    def is_global_layer(self, i):
        """Layer i uses global attention iff i % global_attn_every_n_layers == 0, else the 2*(local_attention//2)+1 band."""
        return i % self.global_attn_every_n_layers == 0

    @property
    def half_window(self):
        """|q - kv| <= half_window in local layers (TF:masking_utils.py:141-151 with sliding_window = local_attention // 2)."""
        return self.local_attention // 2

    def _apply(self, values, tokens):
        for k, v in values.items():
            setattr(self, k, v)
        for k, v in tokens.items():
            setattr(self, k, v)

    def to_dict(self):
        out = super().to_dict()
        out.pop("reference_compile", None)  # ref:cm3p/configuration_cm3p.py:87-90
        return out


# Every concrete class defines its own __init__: transformers 5.x turns config subclasses into dataclasses and only
# leaves a hand-written __init__ alone; 4.55 (the reference's pin) does not care either way.
class CM3PMetadataConfig(_EncoderRules, PretrainedConfig):
    model_type = "CM3PMetadata"
    base_config_key = "metadata_config"

    def __init__(self, **kwargs):
        values, tokens = _split_kwargs(_METADATA_FIELDS, _TOKEN_IDS, kwargs)
        super().__init__(**tokens, **kwargs)
        self._apply(values, tokens)


class CM3PAudioConfig(_EncoderRules, PretrainedConfig):
    model_type = "CM3PAudio"
    base_config_key = "audio_config"

    def __init__(self, **kwargs):
        kwargs.pop("vocab_size", None)
        values, _ = _split_kwargs(_AUDIO_FIELDS, {}, kwargs)
        super().__init__(**kwargs)
        self._apply(values, {})
        self.vocab_size = 1  # the audio tower is fed inputs_embeds; its embedding table is a 1-row placeholder
        if not hasattr(self, "pad_token_id"):
            self.pad_token_id = None


class CM3PBeatmapConfig(_EncoderRules, PretrainedConfig):
    model_type = "CM3PBeatmap"
    is_composition = True
    base_config_key = "beatmap_config"
    sub_configs = {"audio_config": CM3PAudioConfig}

    def __init__(self, audio_config=None, attn_implementation=None, **kwargs):
        values, tokens = _split_kwargs(_BEATMAP_FIELDS, _TOKEN_IDS, kwargs)
        super().__init__(attn_implementation=attn_implementation, **tokens, **kwargs)
        self._apply(values, tokens)
        if isinstance(audio_config, CM3PAudioConfig):
            self.audio_config = audio_config
        else:
            audio_config = dict(audio_config or {})
            audio_config.pop("model_type", None)
            self.audio_config = CM3PAudioConfig(attn_implementation=attn_implementation, **audio_config)


class CM3PConfig(PretrainedConfig):
    model_type = "CM3P"
    is_composition = True
    sub_configs = {"metadata_config": CM3PMetadataConfig, "beatmap_config": CM3PBeatmapConfig}

    def __init__(self, metadata_config=None, beatmap_config=None, projection_dim=512, logit_scale_init_value=2.6592,
                 initializer_factor=1.0, initializer_range=0.02, loss_type=None, has_decoder_head=False,
                 attn_implementation=None, **kwargs):
        super().__init__(attn_implementation=attn_implementation, **kwargs)

        def build(cls, value):
            if isinstance(value, cls):
                return value
            value = dict(value or {})
            value.pop("model_type", None)
            return cls(attn_implementation=attn_implementation, **value)

        self.metadata_config = build(CM3PMetadataConfig, metadata_config)
        self.beatmap_config = build(CM3PBeatmapConfig, beatmap_config)
        self.projection_dim = projection_dim
        self.logit_scale_init_value = logit_scale_init_value
        self.initializer_factor = initializer_factor
        self.initializer_range = initializer_range
        self.loss_type = loss_type
        self.has_decoder_head = has_decoder_head


def _register():
    for cls in (CM3PMetadataConfig, CM3PAudioConfig, CM3PBeatmapConfig, CM3PConfig):
        try:
            AutoConfig.register(cls.model_type, cls)
        except ValueError:
            pass  # already registered (e.g. the reference package was imported first)


_register()

__all__ = ["CM3PConfig", "CM3PMetadataConfig", "CM3PAudioConfig", "CM3PBeatmapConfig"]
