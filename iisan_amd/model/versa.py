"""IISAN-Versa wrapper mirroring `Code_Cached_Asym/model/model.py:255-429`: text and image towers of different depth and
width on cached CLS taps, with group layer-drop and dim-align.  Same constructor, forward signature, return convention and
state-dict keys as the reference; the forward is one fused HIP call (`iisan_side_net_fwd`, versa mode)."""
import torch
import torch.nn as nn

from .. import ops
from .modules import AdapterBlock

__all__ = ["VersaIISANAdaptedMModel"]


class VersaIISANAdaptedMModel(nn.Module):
    cached = True

    def __init__(self, mm_model, args):
        super().__init__()
        E, Dt, Di = args.embedding_dim, args.text_embedding_dim, args.image_embedding_dim
        self.cv_pre_fc = nn.Linear(E, E)                               # model.py:261-262
        self.bert_pre_fc = nn.Linear(E, E)
        if args.remove_first == "TRUE":                                # model.py:263-268
            self.side_bert_adapter_num_list = [int(i) + 1 for i in args.side_adapter_bert_list.split(",")]
            self.side_cv_adapter_num_list = [int(i) + 1 for i in args.side_adapter_vit_list.split(",")]
        else:
            self.side_bert_adapter_num_list = [0] + [int(i) + 1 for i in args.side_adapter_bert_list.split(",")]
            self.side_cv_adapter_num_list = [0] + [int(i) + 1 for i in args.side_adapter_vit_list.split(",")]
        if "inter" not in args.modality:
            raise NotImplementedError(f"modality {args.modality!r}: the IISAN wrapper serves 'intra_inter' and 'inter'")
        if args.cv_adapter_down_size != args.bert_adapter_down_size:
            raise NotImplementedError("the HIP side network uses one bottleneck width for all towers")
        n_cv, n_t = len(self.side_cv_adapter_num_list), len(self.side_bert_adapter_num_list)
        drop = args.adapter_dropout_rate
        self.cv_adapter_list = nn.ModuleList([AdapterBlock(args, Di, args.cv_adapter_down_size, drop) for _ in range(n_cv)])
        self.bert_adapter_list = nn.ModuleList([AdapterBlock(args, Dt, args.bert_adapter_down_size, drop) for _ in range(n_t)])
        if Dt > Di:                                                    # model.py:277-285
            self.down_project_list = nn.ModuleList([nn.Linear(Dt, Di) for _ in range(n_cv)])
            self.mm_adapter_list = nn.ModuleList([AdapterBlock(args, Di, args.cv_adapter_down_size, drop) for _ in range(n_cv)])
        elif Dt < Di:
            self.down_project_list = nn.ModuleList([nn.Linear(Di, Dt) for _ in range(n_t)])
            self.mm_adapter_list = nn.ModuleList([AdapterBlock(args, Dt, args.bert_adapter_down_size, drop) for _ in range(n_t)])
        else:
            self.mm_adapter_list = nn.ModuleList([AdapterBlock(args, Dt, args.bert_adapter_down_size, drop) for _ in range(n_t)])
        self.fc_bert = nn.Linear(Dt, E)                                # model.py:290-299
        self.fc_cv = nn.Linear(Di, E)
        d = min(Di, Dt)
        self.fc_mm = nn.Linear(d, d)
        self.fc_mm_down = nn.Linear(d, E)
        self.gated = args.fusion_method == "gated"
        if self.gated:                                                 # model.py:300-320
            self.side_gate_params_text = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(n_t)])
            self.side_gate_params_cv = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(n_cv)])
            self.side_gate_params_mm = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(min(n_cv, n_t))])
        self.args = args
        self.n_cv, self.n_t, self.align = n_cv, n_t, Di != Dt
        self.remove_first = args.remove_first == "TRUE"
        self._order = ops.versa_param_order(n_cv, n_t, self.align)

    def _abi_params(self, device):
        sd = dict(self.named_parameters())
        out = []
        for k in self._order:
            if k in sd:
                out.append(sd[k])
            else:
                assert "side_gate" in k, k
                out.append(torch.zeros(1, device=device))
        return out

    def forward_item3(self, sample_items_images, sample_items_text):
        a = self.args
        tc = sample_items_images.reshape(-1, sample_items_images.shape[-2], a.image_embedding_dim).float().contiguous()
        tt = sample_items_text.reshape(-1, sample_items_text.shape[-2], a.text_embedding_dim).float().contiguous()   # model.py:402
        cfg = ops.make_versa_cfg(a.image_embedding_dim, a.text_embedding_dim, a.cv_adapter_down_size, a.embedding_dim, self.gated,
                                 a.adapter_activation == "GELU", self.remove_first, tc.shape[1], tt.shape[1],
                                 self.side_cv_adapter_num_list, self.side_bert_adapter_num_list,
                                 taps_exact16=sample_items_images.dtype == torch.float16 and sample_items_text.dtype == torch.float16)
        item3 = ops.SideNetFn.apply(cfg, tc, tt, *self._abi_params(tc.device))
        E = a.embedding_dim
        return item3, (item3[:, :E], [item3[:, E:2 * E], item3[:, 2 * E:]])

    def packed_layers(self):
        """Layer lists a packed `TapStore` must hold for this model, image and text: the towers' tap lists, preceded by
        layer 0 when `remove_first` (the states are seeded with tap 0, model.py:343-351)."""
        pre = [0] if self.remove_first else []
        return pre + list(self.side_cv_adapter_num_list), pre + list(self.side_bert_adapter_num_list)

    def forward_item3_packed(self, taps_cv_sel, taps_text_sel, exact16: bool = False):
        """Taps holding ONLY the layers each tower reads, in `packed_layers()` order (`iisan_amd.tapstore.TapStore.gather`):
        [M, n_img, D_i] and [M, n_text, D_t] instead of the reference's [.., 25, 1024] / [.., 81, 8192] files
        (Code_Cached_Asym/data_utils/dataset.py:37-98).  `exact16`: both stores hold fp16 (the gathered fp32 values are exact
        in fp16)."""
        a = self.args
        o = 1 if self.remove_first else 0
        ni, nt = len(self.side_cv_adapter_num_list), len(self.side_bert_adapter_num_list)
        assert taps_cv_sel.shape[1] == ni + o and taps_text_sel.shape[1] == nt + o, (taps_cv_sel.shape, taps_text_sel.shape)
        cfg = ops.make_versa_cfg(a.image_embedding_dim, a.text_embedding_dim, a.cv_adapter_down_size, a.embedding_dim, self.gated,
                                 a.adapter_activation == "GELU", self.remove_first, ni + o, nt + o,
                                 list(range(o, ni + o)), list(range(o, nt + o)), taps_exact16=exact16)
        item3 = ops.SideNetFn.apply(cfg, taps_cv_sel.float().contiguous(), taps_text_sel.float().contiguous(),
                                    *self._abi_params(taps_cv_sel.device))
        E = a.embedding_dim
        return item3, (item3[:, :E], [item3[:, E:2 * E], item3[:, 2 * E:]])

    def forward(self, sample_items_images, sample_items_text):
        return self.forward_item3(sample_items_images, sample_items_text)[1]
