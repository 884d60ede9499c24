"""Encoders mirroring `Code_Uncached/model/encoders.py`: `MM_Encoder`, `Vit_Encoder`, `Bert_Encoder`,
`Text_Encoder`, `User_Encoder` — same constructors, attribute paths and state-dict keys.

The frozen ViT/BERT are accepted as HuggingFace modules (`ViTForImageClassification`, `BertModel`, either
state-dict layout) or as the light `FrozenVit` / `FrozenBert` containers below; their weights are packed once into
kernel layout (`iisan_amd.encoders.PackedVit/PackedBert`) the first time a forward runs on the GPU.
"""
import torch
import torch.nn as nn
from torch.nn.init import constant_, xavier_normal_

from .. import _lib, encoders as enc, ops, weights
from .modules import TransformerEncoder

__all__ = ["MM_Encoder", "Vit_Encoder", "Bert_Encoder", "Text_Encoder", "User_Encoder", "FrozenVit", "FrozenBert"]


class FrozenVit(nn.Module):
    """Minimal stand-in for `ViTForImageClassification`: the frozen canonical weights (a plain dict, NOT part of
    `state_dict()`: a checkpoint of a model built on this container holds the trainable tensors only) + the trainable
    `classifier` head the reference re-creates (`Code_Uncached/run.py:56-61`).  Hand a real HF module to `ModelMM`
    instead when checkpoints must carry the encoder weights under the reference's keys."""

    def __init__(self, w: dict, cfg: weights.VitConfig, embedding_dim: int = 64):
        super().__init__()
        self.cfg = cfg
        self.canonical = {k: v for k, v in w.items()}
        self.classifier = nn.Linear(cfg.hidden, embedding_dim)
        xavier_normal_(self.classifier.weight.data)
        constant_(self.classifier.bias.data, 0)

    def canonical_weights(self):
        return self.canonical, self.cfg


class FrozenBert(nn.Module):
    def __init__(self, w: dict, cfg: weights.BertConfig):
        super().__init__()
        self.cfg = cfg
        self.canonical = {k: v for k, v in w.items()}

    def canonical_weights(self):
        return self.canonical, self.cfg


def _vit_canonical(image_net):
    if hasattr(image_net, "canonical_weights"):
        return image_net.canonical_weights()
    c = image_net.config                                   # HuggingFace module
    cfg = weights.VitConfig(hidden=c.hidden_size, layers=c.num_hidden_layers, heads=c.num_attention_heads,
                            mlp=c.intermediate_size, image=c.image_size, patch=c.patch_size, channels=c.num_channels,
                            eps=c.layer_norm_eps)
    return weights.vit_from_hf({k: v for k, v in image_net.state_dict().items() if not k.startswith("classifier")}), cfg


def _bert_canonical(bert_model):
    if hasattr(bert_model, "canonical_weights"):
        return bert_model.canonical_weights()
    c = bert_model.config
    cfg = weights.BertConfig(hidden=c.hidden_size, layers=c.num_hidden_layers, heads=c.num_attention_heads,
                             mlp=c.intermediate_size, vocab=c.vocab_size, max_pos=c.max_position_embeddings,
                             eps=c.layer_norm_eps)
    return weights.bert_from_hf(bert_model.state_dict()), cfg


def _drop_packed(module, incompatible_keys):
    module._packed = None


class Vit_Encoder(nn.Module):                      # encoders.py:23-31
    def __init__(self, image_net, dtype16: int = _lib.IISAN_F16):
        super().__init__()
        self.image_net = image_net
        self.activate = nn.GELU()
        self.dtype16 = dtype16
        self.chunk_items = 0
        self._packed = None
        # the kernel-layout copy is built lazily from the module's weights: a later load_state_dict must not leave a stale one
        self.register_load_state_dict_post_hook(_drop_packed)

    def packed(self, device) -> enc.PackedVit:
        if self._packed is None or self._packed.device != torch.device(device):
            w, cfg = _vit_canonical(self.image_net)
            self._packed = enc.PackedVit(w, cfg, device, self.dtype16)
        return self._packed

    @property
    def n_layers(self):
        return _vit_canonical(self.image_net)[1].layers if self._packed is None else self._packed.cfg.layers

    def forward_taps(self, item_content, tap_layers):
        """CLS rows of the requested hidden states, [M, len(tap_layers), 768] fp32 (no grad)."""
        with torch.no_grad():
            pk = self.packed(item_content.device)
            pk.full_blocks = getattr(self, "full_blocks", False)      # True: every block on every token, like HF (bench.py)
            return pk.forward_taps(item_content.contiguous(), list(tap_layers), self.chunk_items)

    def forward(self, item_content):
        """Reference signature (`encoders.py:29-31`).  Returns `(None, hidden_states)` where each hidden state is the
        [M,1,768] CLS row (`h[:,0]` is what IISAN reads, `model.py:212`); the dead GELU(classifier(...)) output of
        the reference is not produced."""
        L = self.packed(item_content.device).cfg.layers
        taps = self.forward_taps(item_content, range(L + 1))
        return None, tuple(taps[:, l:l + 1] for l in range(L + 1))


class User_Encoder(nn.Module):                     # encoders.py:44-65
    def __init__(self, item_num, max_seq_len, item_dim, num_attention_heads, dropout, n_layers):
        super().__init__()
        self.transformer_encoder = TransformerEncoder(n_vocab=item_num, n_position=max_seq_len, d_model=item_dim,
                                                      n_heads=num_attention_heads, dropout=dropout, n_layers=n_layers)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        if isinstance(module, nn.Embedding):
            xavier_normal_(module.weight.data)
        elif isinstance(module, nn.Linear):
            xavier_normal_(module.weight.data)
            if module.bias is not None:
                constant_(module.bias.data, 0)

    def forward(self, input_embs, log_mask, local_rank=None):
        return self.transformer_encoder(input_embs, log_mask, None)


class Text_Encoder(nn.Module):                     # encoders.py:68-91
    def __init__(self, bert_model, args, item_embedding_dim, word_embedding_dim, dtype16: int = _lib.IISAN_F16):
        super().__init__()
        self.bert_model = bert_model
        self.fc = nn.Linear(word_embedding_dim, item_embedding_dim)
        self.activate = nn.GELU()
        self.args = args
        self.dtype16 = dtype16
        self.chunk_items = 0
        self._packed = None
        self.register_load_state_dict_post_hook(_drop_packed)

    def packed(self, device) -> enc.PackedBert:
        if self._packed is None or self._packed.device != torch.device(device):
            w, cfg = _bert_canonical(self.bert_model)
            self._packed = enc.PackedBert(w, cfg, device, self.dtype16)
        return self._packed

    def forward_taps(self, text, tap_layers):
        with torch.no_grad():
            pk = self.packed(text.device)
            pk.full_blocks = getattr(self, "full_blocks", False)
            return pk.forward_taps(text.contiguous().to(torch.int64), list(tap_layers), self.chunk_items)

    def forward(self, text):
        L = self.packed(text.device).cfg.layers
        taps = self.forward_taps(text, range(L + 1))
        return None, tuple(taps[:, l:l + 1] for l in range(L + 1))


class Bert_Encoder(nn.Module):                     # encoders.py:116-159
    def __init__(self, args, bert_model):
        super().__init__()
        self.args = args
        self.attributes2length = {'title': args.num_words_title * 2, 'abstract': args.num_words_abstract * 2,
                                  'body': args.num_words_body * 2}
        for key in list(self.attributes2length.keys()):
            if key not in args.news_attributes:
                self.attributes2length[key] = 0
        self.attributes2start = {key: sum(list(self.attributes2length.values())[:list(self.attributes2length.keys()).index(key)])
                                 for key in self.attributes2length.keys()}
        assert len(args.news_attributes) > 0
        self.text_encoders = nn.ModuleDict({'title': Text_Encoder(bert_model, args, args.embedding_dim, args.word_embedding_dim)})
        self.newsname = [name for name in set(args.news_attributes) & {'title', 'abstract', 'body'}]

    def _title(self, news):
        return torch.narrow(news, 1, self.attributes2start['title'], self.attributes2length['title'])

    def forward_taps(self, news, tap_layers):
        return self.text_encoders['title'].forward_taps(self._title(news), tap_layers)

    def forward(self, news):
        return self.text_encoders['title'](self._title(news))


class MM_Encoder(nn.Module):                       # encoders.py:6-21
    def __init__(self, args, image_net, bert_model):
        super().__init__()
        self.cv_encoder = Vit_Encoder(image_net=image_net)
        self.bert_encoder = Bert_Encoder(args=args, bert_model=bert_model)

    def forward(self, sample_items_images, sample_items_text):
        raise NotImplementedError("MM_Encoder without the IISAN wrapper is the reference's full-fine-tuning baseline "
                                  "(encoders.py:17-21), outside the IISAN hot path; wrap it with IISANAdaptedMModel")
