"""Parameter containers mirroring `Code_*/model/modules.py` (identical in the three reference variants).

The sub-modules exist so that `state_dict()` keys, initialisers and the trainer's name-based freezing/grouping
(`Code_Uncached/run.py:218-224,296-321`) are exactly the reference's.  Their arithmetic is not executed module by
module: `User_Encoder` and `IISANAdaptedMModel` hand the whole parameter set to one fused HIP call.
"""
import torch
import torch.nn as nn

from .. import ops

__all__ = ["PositionwiseFeedForward", "SelfAttention", "MultiHeadedAttention", "TransformerBlock",
           "TransformerEncoder", "AdapterBlock"]


class PositionwiseFeedForward(nn.Module):          # modules.py:6-18
    def __init__(self, d_model, d_inner, dropout):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_inner)
        self.w_2 = nn.Linear(d_inner, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout = nn.Dropout(dropout)
        self.activate = nn.ReLU()


class SelfAttention(nn.Module):                    # modules.py:21-32
    def __init__(self, temperature, dropout):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(dropout)


class MultiHeadedAttention(nn.Module):             # modules.py:35-64
    def __init__(self, n_heads, d_model, dropout):
        super().__init__()
        assert d_model % n_heads == 0
        self.d_model, self.d_k, self.n_heads = d_model, d_model // n_heads, n_heads
        self.d_v = self.d_k
        self.w_Q = nn.Linear(d_model, n_heads * self.d_k, bias=False)
        self.w_K = nn.Linear(d_model, n_heads * self.d_k, bias=False)
        self.w_V = nn.Linear(d_model, n_heads * self.d_v, bias=False)
        self.fc = nn.Linear(n_heads * self.d_v, d_model, bias=False)
        self.self_attention = SelfAttention(temperature=self.d_k ** 0.5, dropout=dropout)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


class TransformerBlock(nn.Module):                 # modules.py:67-76
    def __init__(self, d_model, n_heads, d_inner, dropout):
        super().__init__()
        self.multi_head_attention = MultiHeadedAttention(n_heads=n_heads, d_model=d_model, dropout=dropout)
        self.feed_forward = PositionwiseFeedForward(d_model=d_model, d_inner=d_inner, dropout=dropout)


class TransformerEncoder(nn.Module):               # modules.py:79-96
    def __init__(self, n_vocab, n_position, d_model, n_heads, dropout, n_layers):
        super().__init__()
        self.position_embedding = nn.Embedding(n_position, d_model)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(d_model=d_model, n_heads=n_heads, d_inner=d_model * 4, dropout=dropout) for _ in range(n_layers)])
        self.n_position, self.d_model, self.n_heads, self.n_layers, self.p_drop = n_position, d_model, n_heads, n_layers, dropout
        self._order = ops.sasrec_param_order(n_layers)

    def abi_params(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in self._order]

    def forward(self, input_embs, log_mask, att_mask=None):
        """`att_mask` is accepted for signature parity (`modules.py:89`) and ignored: the kernel derives the causal +
        padding mask from `log_mask` exactly as `User_Encoder.forward` builds it (`encoders.py:60-64`)."""
        p = self.p_drop if self.training else 0.0          # nn.Dropout semantics: active in train() only
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0      # CPU generator: no device sync
        cfg = ops.make_sasrec_cfg(input_embs.shape[1], self.d_model, self.n_heads, self.n_layers, p, seed)
        return ops.SasrecFn.apply(cfg, input_embs, log_mask, *self.abi_params())


class AdapterBlock(nn.Module):                     # modules.py:98-116
    def __init__(self, args, input_size, down_size, dropout=0.1):
        super().__init__()
        self.fc_down = nn.Linear(input_size, down_size)
        nn.init.normal_(self.fc_down.weight, std=1e-2)
        nn.init.zeros_(self.fc_down.bias)
        self.gelu = args.adapter_activation == "GELU"
        self.activate = nn.GELU() if self.gelu else nn.ReLU()
        self.fc_up = nn.Linear(down_size, input_size)
        nn.init.normal_(self.fc_up.weight, std=1e-2)
        nn.init.zeros_(self.fc_up.bias)
        self.dropout = nn.Dropout(dropout)          # never applied by the reference either (modules.py:112-116)

    def forward(self, input_embs):
        """Stand-alone use (not on the IISAN hot path, where the 21 blocks run inside one fused side-network call):
        two HIP Linear calls with the activation between them."""
        x = ops.LinearFn.apply(input_embs, self.fc_down.weight, self.fc_down.bias)
        x = self.activate(x)
        return ops.LinearFn.apply(x, self.fc_up.weight, self.fc_up.bias) + input_embs
