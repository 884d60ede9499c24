import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_io as gio, helpers
from oracle import iisan_oracle as O
z, vw, bw, b, P = gio.e2e_small_inputs()
args = helpers.make_args(side_adapter_vit_list="0,1", side_adapter_bert_list="0,1", num_words_title=8)
model = helpers.build_model(args, 40, b.pop_prob, vw, gio.E2E_VIT, bw, gio.E2E_BERT, cached=False)
helpers.load_trainables(model, P); model.eval()
ids, lm, img, txt = b.ids.cuda().view(-1), b.log_mask.cuda(), b.images.cuda(), b.text.cuda()
loss = model(ids, img, txt, lm, 0); loss.backward()
tc = model.mm_encoder.cv_encoder.forward_taps(img, [0,1,2]).cpu(); tt = model.mm_encoder.bert_encoder.forward_taps(txt, [0,1,2]).cpu()
with torch.no_grad():
    oc = O.vit_cls_taps(b.images, vw, gio.E2E_VIT); ot = O.bert_cls_taps(b.text, bw, gio.E2E_BERT)
print("tap rel err cv", [((tc[:,l]-oc[:,l]).norm()/oc[:,l].norm()).item() for l in range(3)], "text", [((tt[:,l]-ot[:,l]).norm()/ot[:,l].norm()).item() for l in range(3)])
Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
lo, _ = O.model_loss_from_taps(b.ids, tc, tt, b.log_mask, b.pop_prob, Pg, O.side_layer_list("0,1", False)); lo.backward()
print("loss hip", loss.item(), "oracle-on-hip-taps", lo.item(), "golden", float(z["loss"]))
rows = []
for n, p in model.named_parameters():
    if p.requires_grad:
        g = p.grad.cpu().double(); go = Pg[n].grad.double()
        ref = torch.from_numpy(z["g/" + n]).double(); gs = torch.from_numpy(gio.sample_like_golden(p.grad)).double()
        rows.append((((gs - ref).norm() / (ref.norm() + 1e-12)).item(), ((g - go).norm() / (go.norm() + 1e-12)).item(), n))
rows.sort(reverse=True)
for r in rows[:12]: print(f"vs golden {r[0]:.3e}   vs oracle-on-same-taps {r[1]:.3e}   {r[2]}")
