"""Development aid: profiles/rNN_roofline_traffic.json from the two per-kernel counter summaries
(scripts/pmc_summary.py output of a FETCH_SIZE pass and of a WRITE_SIZE pass over scripts/k1_only.py).
usage: traffic_json.py pmc_fetch.txt pmc_write.txt > roofline_traffic.json
Counter unit KB (x 1024); FETCH_SIZE doubled: on gfx950 it reports half the bytes of a 16-byte-per-lane coalesced
streaming read."""
import json, re, sys


def table(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*\S)\s+(\d+)\s*$", line)
        if m:
            out[m.group(1).strip()] = int(m.group(2)) * 1024
    return out


def pick(tab, *needles):
    hits = [v for k, v in tab.items() if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, [k for k in tab if needles[0] in k])
    return hits[0]


F, Wr = table(sys.argv[1]), table(sys.argv[2])
rows, L, n = 250_000_000 // 32, 32, 250_000_000
d = {"how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) -- python3 scripts/k1_only.py on MI355X "
            "(scripts/collect_profiles.sh); counter unit KB (x1024); FETCH_SIZE doubled (gfx950 correction for 16-B/lane coalesced streaming reads); "
            "mean over the 5 launches of each kernel (pmc_fetch / pmc_write files of the same round)",
     "rows_per_launch": rows, "L": L}
for key, needles, alg in (("synth_fwd", ("synth_fwd_kernel<8, 2, false, false, false",), rows * 132),
                          ("synth_fwd_codes_only", ("synth_fwd_kernel<8, 2, true, true, true",), rows * 133),
                          ("synth_bwd", ("synth_bwd_kernel<8, 2, false, false, 4",), rows * 260),
                          ("synth_bwd_adam", ("synth_bwd_kernel<8, 2, false, true",), rows * 772),
                          ("synth_fwd_exact_W_codes", ("synth_fwd_kernel<8, 2, true, true, false",), rows * 137),
                          ("synth_fwd_L31", ("synth_fwd_kernel<8, 3, false, false, false",), rows * 128),
                          ("synth_bwd_L31_no_dq", ("synth_bwd_kernel<8, 3, false, false, 8, false",), rows * 252),
                          ("synth_bwd_L31_dq", ("synth_bwd_kernel<8, 3, false, false, 8, true",), rows * 252)):
    f, w = pick(F, *needles), pick(Wr, *needles)
    d[key + "_FETCH_SIZE_raw_bytes"], d[key + "_WRITE_SIZE_raw_bytes"] = f, w
    d[key + "_bytes_per_launch"], d[key + "_algorithmic_bytes"] = 2 * f + w, alg
d["tern_elements"] = n
tot = 0
for key, needle, alg in (("tern_absstat", "tern_absstat_kernel", 4 * n), ("tern_emit", "tern_emit_kernel", 5 * n)):
    b = 2 * pick(F, needle) + pick(Wr, needle)
    d[key + "_bytes_per_launch"], d[key + "_algorithmic_bytes"] = b, alg
    tot += b
for needle in ("tern_level_kernel<2", "tern_level_kernel<3", "tern_delta_kernel", "tern_codebook_kernel"):
    tot += 2 * pick(F, needle) + pick(Wr, needle)
d["tern_quantise_bytes_per_call"], d["tern_quantise_algorithmic_bytes"] = tot, 9 * n
print(json.dumps(d, indent=1))
