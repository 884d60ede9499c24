#!/usr/bin/env python3
"""Copy one round's evidence (gpurun_out/ev_<tag>/, written on the GPU box by tools/evidence_<tag>.sh) into the tracked summaries under profiles/:
    python tools/assemble_profiles.py r6 ["note on what changed"]
kernel traces -> <tag>_bench / _cached / _versa / _eval _kernel_stats.md; the PMC passes -> <tag>_pmc_traffic.md + profiles/pmc_traffic*.json (read by bench.py);
matrix-pipe counters -> <tag>_mfma_util.md."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
note = sys.argv[2] if len(sys.argv) > 2 else ""
ev = os.path.join(ROOT, "gpurun_out", f"ev_{tag}")
prof = os.path.join(ROOT, "profiles")
rd = lambda n: open(os.path.join(ev, n)).read() if os.path.exists(os.path.join(ev, n)) else ""
R = tag[1:]


def line_of(name):
    t = rd(name).strip()
    return json.loads(t) if t.startswith("{") else None


titles = {"head": ("bench", "the headline (Uncached IISAN, bs = 128, fp16, every block on every token; towers back to back: `--no-overlap-towers`, the profiling form)"),
          "cached": ("cached", "BASELINE config 3: Code_Cached IISAN, bs = 1024 (11,264 item slots), fp32 tap store"),
          "versa": ("versa", "BASELINE config 5 shapes on one GPU: IISAN-Versa, bs = 128, fp16 tap stores (ViT-L 25 x 1024 + Llama-3-70B 81 x 8192)"),
          "eval": ("eval", "the eval path at Scientific size (`bench.py --eval`)")}
for cfg, (out, title) in titles.items():
    d = line_of(f"{cfg}_bench_line.json")
    hdr = f"# Round {R} — rocprofv3 --kernel-trace of {title}\n\nCommand (`tools/evidence_{tag}.sh {tag} trace`)"
    if d:
        rf = d.get("roofline", {})
        hdr += f"; bench line of the same (profiled) run: {d['value']:.0f} {d['unit']}, {d['ms_per_step']:.3f} ms/step; `roofline.frac` {rf.get('frac', 0):.4f} ({rf.get('bound')})"
        if rf.get("traffic") is not None:
            hdr += f"; PMC traffic {rf['traffic'] / 1e9:.2f} GB vs {rf.get('traffic_algorithmic', 0) / 1e6:.0f} MB algorithmic"
        if cfg == "eval" and "recommend_topk" in d.get("config", {}):
            t = d["config"]["recommend_topk"]
            hdr += f"; recommend_topk {t['users_per_s']:.0f} users/s end to end, `iisan_score_topk` alone {t['score_topk_alone']['ms_per_call']:.3f} ms per call"
    if note:
        hdr += "\n" + note
    open(os.path.join(prof, f"{tag}_{out}_kernel_stats.md"), "w").write(hdr + "\n\n" + rd(f"{cfg}_kernel_stats.md"))

cmd = f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-check --no-overlap-towers (two passes, round {R} code, tools/evidence_{tag}.sh {tag} pmc)"
parts = [f"# Round {R} — memory-side traffic (rocprofv3 PMC, 1 x MI355X)\n\nTwo passes per configuration (`FETCH_SIZE` and `WRITE_SIZE` do not fit the TCC counter slots of one pass on gfx950), counters only beside "
         f"`--kernel-trace` (`tools/evidence_{tag}.sh {tag} pmc`, summarised by `tools/pmc_traffic.py`; bytes = 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md):\n\n    {cmd}\n"]
for name, what in (("pmc_head_gemm16.md", "headline: the encoder GEMM launches (`roofline.traffic` of the bench line)"), ("pmc_head_attn.md", "headline: attention"),
                   ("pmc_head_ln.md", "headline: LayerNorm (BERT tower, embeddings)"), ("pmc_head_finalize.md", "headline: stream statistics"), ("pmc_head_fold.md", "headline: LayerNorm weight folding"),
                   ("pmc_cached.md", "Cached bs = 1024: whole step"), ("pmc_versa.md", "Versa bs = 128: whole step"), ("pmc_eval_score_rank.md", "eval: iisan_score_rank"),
                   ("pmc_eval_score_topk.md", "eval: iisan_score_topk"), ("pmc_eval_gemm32.md", "eval: item table products")):
    t = rd(name)
    if t:
        parts.append(f"\n## {what}\n\n{t}")
open(os.path.join(prof, f"{tag}_pmc_traffic.md"), "w").write("".join(parts))


def last_json(name):
    for l in reversed(rd(name).strip().split("\n")):
        if l.startswith("{"):
            return json.loads(l)
    return None


j = last_json("pmc_head_gemm16.md")
if j:
    j.update({"encoder_blocks": "all tokens in every block", "ln_fold": 2, "command": cmd, "kernels": "gemm16*",
              "unit": "bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md)"})
    json.dump(j, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
j = last_json("pmc_eval_score_rank.md")
if j:
    j.update({"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --eval (two passes, round {R} code)", "kernels": "score_rank_mfma_kernel"})
    json.dump(j, open(os.path.join(prof, "pmc_traffic_eval.json"), "w"), indent=1)
t = rd("pmc_traffic_cached.json")
if t:
    open(os.path.join(prof, "pmc_traffic_cached.json"), "w").write(t)

parts = [f"# Round {R} — matrix-pipe utilisation and wave-time split (rocprofv3 PMC, 1 x MI355X)\n\nOne counter pass beside `--kernel-trace` (`tools/evidence_{tag}.sh {tag} mfma`, summarised by `tools/mfma_util.py`; "
         "column definitions in `r4_mfma_util.md`):\n\n    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python3 bench.py <configuration>\n"]
for name, what in (("mfma_head.md", "headline"), ("mfma_cached.md", "Cached bs = 1024"), ("mfma_eval.md", "eval path")):
    t = rd(name)
    if t:
        parts.append(f"\n## {what}\n\n{t}")
open(os.path.join(prof, f"{tag}_mfma_util.md"), "w").write("".join(parts))
print("written:", sorted(f for f in os.listdir(prof) if f.startswith(tag + "_") or f.startswith("pmc_traffic")))
