"""Builders the tests, `bench.py` and `__graft_entry__.smoke()` share: reference-style `args` and a product `ModelMM`
wired the way `Code_Uncached/run.py:161-224` wires the reference (construct, freeze, wrap with the side network,
re-enable by name)."""
from types import SimpleNamespace

from . import weights


def make_args(**kw):
    """The `args` namespace fields the `model/` package reads, at the values of the reference's IISAN
    launcher (`Code_Uncached/script/run_IISAN.py`); keyword arguments override."""
    a = dict(max_seq_len=10, l2_weight=0.1, embedding_dim=64, num_attention_heads=2, drop_rate=0.1,
             transformer_block=2, modality="intra_inter", CV_model_load="vit-base-mae", bert_model_load="bert_base_uncased",
             word_embedding_dim=768, num_words_title=30, num_words_abstract=0, num_words_body=0,
             news_attributes=["title"], remove_first="None", side_adapter_vit_list="1,3,5,7,9,11",
             side_adapter_bert_list="1,3,5,7,9,11", cv_adapter_down_size=64, bert_adapter_down_size=64,
             adapter_dropout_rate=0.1, adapter_activation="RELU", fusion_method="gated",
             lr=2e-4, adapter_cv_lr=1e-4, adapter_bert_lr=1e-4, fine_tune_lr_image=1e-4, fine_tune_lr_text=5e-5)
    a.update(kw)
    return SimpleNamespace(**a)


def build_model(args, item_num, pop, vit_w=None, vit_cfg=None, bert_w=None, bert_cfg=None, cached=False, device="cuda"):
    """Product `ModelMM` on canonical encoder weights (`iisan_amd.weights`); `cached`: False | True | "versa"."""
    from . import trainer
    from .model import FrozenBert, FrozenVit, ModelMM
    vit = FrozenVit(vit_w or {}, vit_cfg or weights.VIT_BASE, args.embedding_dim)
    bert = FrozenBert(bert_w or {}, bert_cfg or weights.BERT_BASE)
    model = ModelMM(args, item_num, True, vit, bert, pop)
    trainer.apply_iisan_freeze_rules(model, args, cached=cached)
    return model.to(device)


def load_trainables(model, P):
    """Copy a `{state-dict key: tensor}` set of trainable tensors into the model; unknown keys are an error."""
    missing, unexpected = model.load_state_dict({k: v for k, v in P.items()}, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if m in P], missing
