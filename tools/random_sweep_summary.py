"""Summary of a randomised sweep (SLS_TEST_EXTRA_SEEDS=n pytest -k randomised) from gpurun_out/test_evidence.json ->
gpurun_out/r03/random_sweep.json: how many cases hold at the flat 1e-6, which needed the conditioning term, largest errors."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
e = json.load(open(os.path.join(R, "gpurun_out", "test_evidence.json")))
r = [x for x in e if x.get("kind") == "randomised"]
need = [x for x in r if not x["flat_1e6"]]
flat = [x for x in r if x["flat_1e6"]]
out = dict(cases=len(r), seeds=max(x["seed"] for x in r) + 1, paths=sorted(set(x["path"] for x in r)), flat_1e6=len(flat),
           max_err_of_the_flat_cases=dict(sigma=max(x["sigma"] for x in flat), acq=max(max(x["acq0"], x["acq1"]) for x in flat),
                                          grad=max(max(x["grad0"], x["grad1"]) for x in flat)),
           needed_conditioning_term=[{k: x[k] for k in ("seed", "path", "N", "D", "b", "kappa", "sigma_min", "sigma", "bound_sigma_rel", "acq0",
                                                        "acq1", "grad0", "grad1")} for x in need])
os.makedirs(os.path.join(R, "gpurun_out", "r03"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "r03", "random_sweep.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "needed_conditioning_term"}), len(need), "cases needed the conditioning term")
