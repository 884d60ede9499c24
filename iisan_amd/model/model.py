"""`ModelMM` and `IISANAdaptedMModel` mirroring `Code_Uncached/model/model.py` (and the Cached wrapper of
`Code_Cached/model/model.py:257-349`): identical constructors, forward signatures, return conventions and state-dict
keys; the forward runs on the HIP library (frozen encoders -> CLS taps -> fused side network -> com_dense -> SASRec ->
fused in-batch CE)."""
import torch
import torch.nn as nn

from .. import ops
from .encoders import MM_Encoder, User_Encoder
from .modules import AdapterBlock

__all__ = ["ModelMM", "IISANAdaptedMModel", "CachedIISANAdaptedMModel"]


class _SideNetBase(nn.Module):
    """Everything the Uncached and Cached IISAN wrappers share (`model.py:166-205`)."""

    cached = False

    def _build(self, args):
        embedding_dim = 768
        if args.remove_first == "TRUE":             # model.py:172-174 (the reference swaps the two list names here)
            self.side_bert_adapter_num_list = [int(i) + 1 for i in args.side_adapter_vit_list.split(",")]
            self.side_cv_adapter_num_list = [int(i) + 1 for i in args.side_adapter_bert_list.split(",")]
        else:                                       # model.py:176-177
            self.side_bert_adapter_num_list = [0] + [int(i) + 1 for i in args.side_adapter_vit_list.split(",")]
            self.side_cv_adapter_num_list = [0] + [int(i) + 1 for i in args.side_adapter_vit_list.split(",")]
        if "inter" not in args.modality:
            # the reference wrapper returns (cv, [text, mm]) whatever the modality (model.py:271) and ModelMM's plain 'intra'
            # branch concatenates cv with that LIST (model.py:73-74): it does not run there either
            raise NotImplementedError(f"modality {args.modality!r}: the IISAN wrapper serves 'intra_inter' and 'inter'")
        if self.side_bert_adapter_num_list != self.side_cv_adapter_num_list:
            raise NotImplementedError("different tap lists per tower are the Versa variant (Code_Cached_Asym)")
        n = len(self.side_cv_adapter_num_list)
        # model.py:178-205: the intra-modal towers (cv, text) exist with "intra" in the modality, the inter-modal tower (mm)
        # with "inter".  With modality "inter" only the mm tower has parameters (and state-dict keys); the fused HIP side
        # network still runs three towers — the other two on zero placeholders, their outputs unused by ModelMM.
        self.intra = "intra" in args.modality
        if self.intra:
            self.cv_adapter_list = nn.ModuleList([AdapterBlock(args, embedding_dim, args.cv_adapter_down_size, args.adapter_dropout_rate) for _ in range(n)])
            self.bert_adapter_list = nn.ModuleList([AdapterBlock(args, args.word_embedding_dim, args.bert_adapter_down_size, args.adapter_dropout_rate) for _ in range(n)])
        self.mm_adapter_list = nn.ModuleList([AdapterBlock(args, args.word_embedding_dim, args.bert_adapter_down_size, args.adapter_dropout_rate) for _ in range(n)])
        if self.intra:
            self.fc_bert = nn.Linear(embedding_dim, embedding_dim)
            self.fc_cv = nn.Linear(embedding_dim, embedding_dim)
        self.fc_mm = nn.Linear(args.word_embedding_dim, args.word_embedding_dim)
        self.fc_mm_down = nn.Linear(args.word_embedding_dim, args.embedding_dim)
        self.gated = args.fusion_method == "gated"
        if self.gated:                              # model.py:190-205
            if self.intra:
                self.side_gate_params_text = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(n)])
                self.side_gate_params_cv = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(n)])
            self.side_gate_params_mm = nn.ParameterList([nn.Parameter(torch.ones(1) * 0) for _ in range(n)])
        self.args = args
        self.n_side = n
        self.remove_first = args.remove_first == "TRUE"
        self._order = ops.side_param_order(n, cached=self.cached)
        self._placeholders = {}

    def _abi_params(self, device):
        sd = dict(self.named_parameters())
        out = []
        for k in self._order:
            if k in sd:
                out.append(sd[k])
                continue
            # gates do not exist when fusion_method != "gated"; the cv / text towers do not exist with modality "inter"
            assert "side_gate" in k or (not self.intra and any(t in k for t in ("cv_adapter_list", "bert_adapter_list", "fc_cv", "fc_bert"))), k
            ph = self._placeholders.get(k)
            if ph is None or ph.device != device:
                shape = (1,) if "side_gate" in k else self._placeholder_shape(k)
                ph = self._placeholders[k] = torch.zeros(shape, device=device)
            out.append(ph)
        return out

    def _placeholder_shape(self, k):
        D, r = self.fc_mm.in_features, self.mm_adapter_list[0].fc_down.out_features
        if "fc_down.weight" in k: return (r, D)
        if "fc_down.bias" in k: return (r,)
        if "fc_up.weight" in k: return (D, r)
        if "fc_up.bias" in k: return (D,)
        return (D, D) if k.endswith("weight") else (D,)            # fc_cv / fc_bert

    def _side(self, taps_cv, taps_text, tap_index, first_index):
        cfg = ops.make_side_cfg(self.n_side, taps_cv.shape[-1], self.mm_adapter_list[0].fc_down.out_features,
                                self.fc_mm_down.out_features, self.gated, self.mm_adapter_list[0].gelu, self.remove_first,
                                taps_cv.shape[1], taps_text.shape[1], tap_index, first_index)
        item3 = ops.SideNetFn.apply(cfg, taps_cv, taps_text, *self._abi_params(taps_cv.device))
        E = cfg.emb
        return item3, (item3[:, :E], [item3[:, E:2 * E], item3[:, 2 * E:]])


class IISANAdaptedMModel(_SideNetBase):
    """Uncached wrapper (`Code_Uncached/model/model.py:166-271`): owns the two frozen encoders."""

    overlap_towers = True       # scheduling only (forward_item3): the two frozen towers on two HIP streams

    def __init__(self, mm_model, args):
        super().__init__()
        self.cv_encoder = mm_model.cv_encoder
        self.bert_encoder = mm_model.bert_encoder
        self._build(args)

    def forward_item3(self, sample_items_images, sample_items_text, item_ids=None):
        layers = self.side_cv_adapter_num_list
        need = ([0] if self.remove_first else []) + list(layers)        # model.py:215-218 seeds states with tap 0
        need = sorted(set(need))
        if item_ids is not None:
            # encode every distinct item of the batch once.  Item content is a function of the item id
            # (`Build_MM_Dataset.__getitem__`, dataset.py:73-84; id 0 = the all-zero padding content), encoder rows are
            # independent and bit-reproducible, so scattering the unique taps back gives exactly the [M, ...] taps the
            # reference computes redundantly (57 % of Scientific-shaped slots are padding, ~11 % of the rest repeats).
            ids = item_ids.reshape(-1)
            uniq, inverse = torch.unique(ids, return_inverse=True)
            first = torch.full_like(uniq, ids.numel()).scatter_reduce_(0, inverse, torch.arange(ids.numel(), device=ids.device),
                                                                        reduce="amin")
            taps_cv = self.cv_encoder.forward_taps(sample_items_images.index_select(0, first), need).index_select(0, inverse)
            taps_text = self.bert_encoder.forward_taps(sample_items_text.index_select(0, first), need).index_select(0, inverse)
        else:
            if self.overlap_towers and sample_items_images.is_cuda:
                # the two towers on two HIP streams of their own — the image tower on a HIGH-priority stream, the text tower on a
                # normal one — so that the text tower's kernels only fill what the image tower's persistent GEMMs leave free (their
                # partial last rounds).  Same kernels, same results (tests/test_gpu_trainable.py: bit-identical embeddings).
                # profiles/r5_overlap.md (same box, three interleaved rounds of 20 steps): -0.33 / -0.41 ms per step, every round; with
                # BOTH towers at normal priority the step is bimodal (-0.5 ms or +1.2 .. +4.3 ms: a text-tower GEMM that wins a CU keeps
                # it for its whole static tile list and the image tower's kernel waits for it).  Round 6: the default
                # (`overlap_towers = False` / `bench.py --no-overlap-towers` for profiling runs: per-kernel durations of an overlapped
                # run — HIP events and rocprofv3 alike — are inflated by the sharing, profiles/r4_overlap_by_stream.md).
                cur = torch.cuda.current_stream()
                hi, side = self.tower_streams()
                side.wait_stream(cur)
                hi.wait_stream(cur)
                with torch.cuda.stream(side):
                    taps_text = self.bert_encoder.forward_taps(sample_items_text, need)
                with torch.cuda.stream(hi):
                    taps_cv = self.cv_encoder.forward_taps(sample_items_images, need)
                cur.wait_stream(side)
                cur.wait_stream(hi)
                taps_text.record_stream(cur)
                taps_cv.record_stream(cur)
            else:
                taps_cv = self.cv_encoder.forward_taps(sample_items_images, need)
                taps_text = self.bert_encoder.forward_taps(sample_items_text, need)
        return self._side(taps_cv, taps_text, [need.index(l) for l in layers], need.index(0) if self.remove_first else 0)

    def tower_streams(self):
        """(image-tower stream: high priority, text-tower stream: normal priority) of the `overlap_towers` mode, created on first use."""
        if getattr(self, "_tower_streams", None) is None:
            self._tower_streams = (torch.cuda.Stream(priority=-1), torch.cuda.Stream())
        return self._tower_streams

    def forward(self, sample_items_images, sample_items_text):
        return self.forward_item3(sample_items_images, sample_items_text)[1]


class CachedIISANAdaptedMModel(_SideNetBase):
    """Cached wrapper (`Code_Cached/model/model.py:257-349`): inputs are the precomputed CLS taps
    `[bs, S+1, L+1, 768]` (train) or `[B, L+1, 768]` (eval); only the two 768->64 heads of the encoders are kept."""

    cached = True

    def __init__(self, mm_model, args):
        super().__init__()
        self.cv_pre_fc = mm_model.cv_encoder.image_net.classifier
        self.bert_pre_fc = mm_model.bert_encoder.text_encoders.title.fc
        self._build(args)

    def forward_item3(self, sample_items_images, sample_items_text):
        tc = sample_items_images.reshape(-1, sample_items_images.shape[-2], sample_items_images.shape[-1])
        tt = sample_items_text.reshape(-1, sample_items_text.shape[-2], sample_items_text.shape[-1])
        return self._side(tc.contiguous(), tt.contiguous(), list(self.side_cv_adapter_num_list), 0)

    def forward(self, sample_items_images, sample_items_text):
        return self.forward_item3(sample_items_images, sample_items_text)[1]

    def packed_layers(self):
        """Layer list a packed `TapStore` must hold for this model (both modalities): the side network's tap list,
        preceded by layer 0 when `remove_first` (the states are seeded with tap 0, model.py:215-218)."""
        return ([0] if self.remove_first else []) + list(self.side_cv_adapter_num_list)

    def forward_item3_packed(self, taps_cv_sel, taps_text_sel, exact16: bool = False):
        """Taps that hold ONLY the layers this side network reads, in `packed_layers()` order
        (`iisan_amd.tapstore.TapStore.gather`): [M, n, 768] per modality instead of the reference's [.., 13, 768]."""
        o = 1 if self.remove_first else 0
        n = len(self.side_cv_adapter_num_list)
        assert taps_cv_sel.shape[1] == n + o and taps_text_sel.shape[1] == n + o, (taps_cv_sel.shape, taps_text_sel.shape, n)
        return self._side(taps_cv_sel.contiguous(), taps_text_sel.contiguous(), list(range(o, n + o)), 0)


class ModelMM(nn.Module):                          # model.py:14-105
    def __init__(self, args, item_num, use_modal, image_net, bert_model, pop_prob_list):
        super().__init__()
        self.args = args
        self.use_modal = use_modal
        self.max_seq_len = args.max_seq_len
        self.l2_weight = args.l2_weight / 2
        self.pop_prob_list = torch.as_tensor(pop_prob_list, dtype=torch.float32)
        self.user_encoder = User_Encoder(item_num=item_num, max_seq_len=args.max_seq_len, item_dim=args.embedding_dim,
                                         num_attention_heads=args.num_attention_heads, dropout=args.drop_rate,
                                         n_layers=args.transformer_block)
        if not use_modal:
            raise NotImplementedError("use_modal=False (ID embeddings) is not the IISAN hot path")
        self.mm_encoder = MM_Encoder(args, image_net, bert_model)
        if "inter" not in args.modality:
            raise NotImplementedError(f"modality {args.modality!r}: 'intra_inter' (IISAN default) and 'inter' are built; plain "
                                      "'intra' does not run with the IISAN wrapper in the reference either (model.py:73-74)")
        # model.py:36-41: 'intra_inter' -> com_dense(cat[cv, text, mm]); 'inter' -> com_dense(mm) alone (the cv / text towers
        # still run, as in the reference, and receive zero gradients)
        self.inter_only = "intra_inter" not in args.modality
        self.com_dense = nn.Linear(args.embedding_dim * (1 if self.inter_only else 3), args.embedding_dim)
        self.criterion = nn.CrossEntropyLoss()
        # opt-in (not reference behaviour): encode each distinct item id of a batch once (padding = id 0), see
        # IISANAdaptedMModel.forward_item3 (Uncached) and score_embs below (Cached with tap stores).  Requires inputs that
        # are a function of the id, as the datasets produce.
        self.dedup_items = False
        # Cached path only: packed device tap stores (iisan_amd.tapstore.TapStore) for image / text taps.
        # When set, forward() ignores the `sample_items_images/text` arguments and gathers the taps by item id.
        self.tap_stores = None

    def score_embs(self, sample_items_images, sample_items_text, sample_items_id=None):
        enc = self.mm_encoder
        if self.tap_stores is not None and getattr(enc, "cached", False) and sample_items_id is not None:
            st_cv, st_tx = self.tap_stores
            ex16 = st_cv.table.dtype == torch.float16 and st_tx.table.dtype == torch.float16     # fp16 stores: exact taps
            if self.dedup_items:
                # opt-in (SURVEY 8f-3 carried over to the Cached path): the side network and com_dense run once per DISTINCT
                # item id of the batch — on Scientific-shaped batches 57 % of the slots are padding (id 0) and ~11 % of the
                # rest repeat — and the [U, E] result is gathered back to the [M, E] slots.  Rows are independent and
                # bit-reproducible, so the loss is bit-identical; gradients differ by summation order only.
                uniq, inverse = torch.unique(sample_items_id.reshape(-1), return_inverse=True)
                pad = (-uniq.numel()) % 64         # whole 64-row tiles for the K = rows weight-gradient products; the extra
                if pad:                            # rows (copies of the first id) are gathered by nobody: zero gradient
                    uniq = torch.cat([uniq, uniq[:1].expand(pad)])
                item3, _ = enc.forward_item3_packed(st_cv.gather(uniq), st_tx.gather(uniq), exact16=ex16)
                return self.fuse_item3(item3).index_select(0, inverse)
            item3, _ = enc.forward_item3_packed(st_cv.gather(sample_items_id), st_tx.gather(sample_items_id), exact16=ex16)
        elif hasattr(enc, "forward_item3"):
            if self.dedup_items and sample_items_id is not None and not getattr(enc, "cached", False):
                item3, _ = enc.forward_item3(sample_items_images, sample_items_text, sample_items_id)
            else:
                item3, _ = enc.forward_item3(sample_items_images, sample_items_text)
        else:
            raise NotImplementedError("mm_encoder must be wrapped by IISANAdaptedMModel (run.py:214-216)")
        return self.fuse_item3(item3)

    def fuse_item3(self, item3):
        """`com_dense` over the towers' outputs `[M, 3E] = cv | text | mm` (model.py:67-72)."""
        if self.inter_only:
            E = self.args.embedding_dim
            item3 = item3[:, 2 * E:].contiguous()
        return ops.LinearFn.apply(item3, self.com_dense.weight, self.com_dense.bias)

    def forward(self, sample_items_id, sample_items_images, sample_items_text, log_mask, local_rank=None):
        if self.pop_prob_list.device != log_mask.device:
            self.pop_prob_list = self.pop_prob_list.to(log_mask.device)                         # model.py:63
        score_embs = self.score_embs(sample_items_images, sample_items_text, sample_items_id)
        E = self.args.embedding_dim
        input_embs = score_embs.view(-1, self.max_seq_len + 1, E)
        prec_vec = self.user_encoder(input_embs[:, :-1, :].contiguous(), log_mask, local_rank)  # model.py:76-77
        return ops.InbatchCeFn.apply(sample_items_id.view(-1), score_embs, prec_vec.reshape(-1, E), log_mask,
                                     self.pop_prob_list)                                        # model.py:81-104
