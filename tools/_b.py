import json
for l in open("gpurun_out/b_alias.log"):
    if l.startswith("{"):
        j=json.loads(l); print(j["value"], j["ms_per_step"], j["roofline"]["whole_step_frac_reference_flops"], j["roofline"]["traffic"])
