#!/usr/bin/env python3
"""The LZ decoders' throughput model ("model first"): what bounds the one-wave-per-chunk decoder on the
headline workload, and what the designs on the table would reach, from MEASURED inputs only:

  * sequences per chunk of the headline workload (counted here from the liblz4-HC streams of the mix: oracle/_ref);
  * wave instructions per launch (profiles/*_pmc.json: SQ_INSTS_VALU / SALU / LDS of the window kernel);
  * throughput against resident waves per CU (profiles/r04_feasibility.json open_chunks_sweep: 4 / 8 / 16 / 28 waves);
  * the request ceiling of the memory system (r04_feasibility.json: 50.9 G random 128-byte requests / s) and the decoder's
    request count (FETCH_SIZE / 64);
  * the workgroup-per-chunk decoder's instruction and latency budget (profiles/r04_team_phases.jsonl, r04_team.jsonl).

A wave64 vector instruction of this kernel's mix holds its SIMD for 4.1 cycles, a scalar one 4.17 -- MEASURED in round 6
(scripts/microbench/valu_issue.hip -> profiles/r06_valu_issue_*.jsonl; the environment's guide says 2 cycles, which holds
for pure streams of add / sub / logic / mov only) -- at 2.3 GHz under load: a launch cannot end before
VALU x 4.1 / (1 024 SIMDs x 2.3 GHz); how close to that a kernel runs at w waves per CU is read off the sweep. Round 6 also
measured the two experiments the model was missing (profiles/r06_far_ablate.jsonl, r06_token_index.json): they are added as
`round6`. Prints the table of DESIGN.md section 6 and writes profiles/r06_decode_model.json.
usage: decode_model.py [--count-sequences]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SCLK = 2.3e9  # measured under load (r06_valu_issue_*.jsonl, the PMC passes)
VALU_CYCLES, SALU_CYCLES = 4.1, 4.17
SIMDS = 1024
CHUNKS = 65536
RAW = CHUNKS * 65536


def count_sequences(sample_mib=16):
    """Sequences per 64 KiB chunk of the headline workload (liblz4 HC-12 on the mix): a token walk over the streams."""
    from nvcomp_amd import datasets
    from oracle import oracle_py as o

    o.build()
    data = datasets.silesia_style(sample_mib << 20, 0)
    chunks = datasets.split_chunks(data)
    _, outs, errs = o.batch_run(o.LZ4_ENC_HC, chunks, [o.lz4_bound(c.size) + 64 for c in chunks], threads=os.cpu_count(), use_ref=True)
    assert errs == 0
    total = 0
    for c in outs:
        b = bytes(c)
        n, pos, seqs = len(b), 0, 0
        while pos < n:
            t = b[pos]
            pos += 1
            ll = t >> 4
            if ll == 15:
                while True:
                    x = b[pos]
                    pos += 1
                    ll += x
                    if x != 255:
                        break
            pos += ll
            seqs += 1
            if pos >= n:
                break
            pos += 2
            if (t & 15) == 15:
                while True:
                    x = b[pos]
                    pos += 1
                    if x != 255:
                        break
        total += seqs
    return total / len(chunks)


def main():
    prof = os.path.join(REPO, "profiles")
    pmc = json.load(open(os.path.join(prof, "r05_final_pmc.json" if os.path.exists(os.path.join(prof, "r05_final_pmc.json")) else "r04_final_pmc.json")))
    feas = json.load(open(os.path.join(prof, "r04_feasibility.json")))
    seq_per_chunk = count_sequences() if "--count-sequences" in sys.argv else 6200.0  # DESIGN.md 3.1 / docs/HISTORY.md (emulator statistics)
    seqs = seq_per_chunk * CHUNKS
    out = {"inputs": {"sequences_per_chunk": round(seq_per_chunk, 1), "chunks": CHUNKS}, "formats": {}}
    sweep = {r["open_chunks_per_cu"]: r for r in feas["open_chunks_sweep"]["rows"]}
    for algo, key, gb in (("lz4", "GBps_4GiB", None), ("snappy", "snappy_GBps_4GiB", None)):
        c = pmc[algo]
        valu, salu, lds = c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"]
        issue_floor_ms = valu * VALU_CYCLES / (SIMDS * SCLK) * 1e3  # every SIMD issuing a vector instruction every 4.1 cycles
        salu_floor_ms = salu * SALU_CYCLES / (SIMDS * SCLK) * 1e3
        rows = []
        for w in (4, 8, 16, 28):
            g = sweep[w][key]
            ms = RAW / (g * 1e9) * 1e3
            rows.append({"waves_per_cu": w, "GBps": g, "ms": round(ms, 3), "valu_busy": round(issue_floor_ms / ms, 3)})
        ceiling = RAW / (issue_floor_ms * 1e-3) / 1e9
        out["formats"][algo] = {
            "valu_per_sequence": round(valu / seqs, 2), "salu_per_sequence": round(salu / seqs, 2), "lds_per_sequence": round(lds / seqs, 2),
            "issue_floor_ms": round(issue_floor_ms, 3), "scalar_issue_floor_ms": round(salu_floor_ms, 3),
            "salu_busy_at_28_waves": round(salu_floor_ms / rows[-1]["ms"], 3),
            "GBps_at_full_vector_issue": round(ceiling, 1), "sweep": rows,
            "valu_per_sequence_for_850_at_0.88_busy": round(valu / seqs * (rows[-1]["GBps"] / 850.0), 2),
        }
    lz4 = out["formats"]["lz4"]
    req = sweep[28]["requests_128B"]
    out["requests"] = {"per_launch": req, "per_sequence": round(req / seqs, 3), "ceiling_G_per_s": 50.9,
                       "GBps_at_the_request_ceiling": round(RAW / (req / 50.9e9) / 1e9, 1)}
    # ---- the designs on the table, costed with the same two numbers ----
    top = lz4["GBps_at_full_vector_issue"]
    busy28 = lz4["sweep"][-1]["valu_busy"]
    team = [json.loads(l) for l in open(os.path.join(prof, "r04_team_phases.jsonl"))][0]
    designs = [
        {"design": "one wave per chunk, 28 waves per CU (shipped)", "vector_instructions_x": 1.0, "valu_busy": busy28,
         "projected_GBps": round(top * busy28, 0), "requests_x": 1.0, "basis": "measured"},
        {"design": "workgroup per chunk, output in LDS, barriers (round 4, common/lz_team.hip.h)", "vector_instructions_x": 2.0,
         "valu_busy": 0.46, "projected_GBps": 175, "requests_x": 0.15,
         "basis": f"measured: 175 GB/s at full load; {team['cycles_per_chunk_per_wave']} wave-cycles a chunk, "
                  f"{round(100 * (team['share']['wait_A'] + team['share']['wait_B'] + team['share']['wait_C'] + team['share']['wait_L']))} % of them in barrier waits"},
        {"design": "roles coupled by LDS queues, no workgroup barriers, 2 chunks per CU in LDS (DESIGN r4 6.1)",
         "vector_instructions_x": 1.5, "valu_busy": 0.85, "projected_GBps": round(top * 0.85 / 1.5, 0), "requests_x": 0.15,
         "basis": "the team's front end (exit tables, speculation, granule counters) costs 2.0 x the one-wave decoder's "
                  "instructions per token; without the barrier phases an estimated 1.5 x stays (speculated windows are still "
                  "built twice in one step of four, readiness is still polled); 0.85 busy = what 28-32 waves reach"},
        {"design": "sliding 24-32 KiB LDS history, 4-5 chunks per CU, one wave each + deep matches from L2",
         "vector_instructions_x": 1.0, "valu_busy": round(sweep[4]["GBps_4GiB"] / top * 5 / 4, 3),
         "projected_GBps": round(sweep[4]["GBps_4GiB"] * 5 / 4, 0), "requests_x": 0.3,
         "basis": "4 waves per CU measured at 153 GB/s WITH every match free (r04_feasibility whole_chunk_lds_window: a lone "
                  "wave needs 0.41 ms a chunk): five waves per CU cannot issue more than 5/4 of that"},
        {"design": "the same with a team of 4-6 waves per chunk", "vector_instructions_x": 1.5, "valu_busy": 0.8,
         "projected_GBps": round(top * 0.8 / 1.5, 0), "requests_x": 0.3, "basis": "as the queue-coupled team above"},
    ]
    out["designs"] = designs
    out["verdict"] = (f"850 GB/s needs <= {lz4['valu_per_sequence_for_850_at_0.88_busy']} vector instructions per sequence at the "
                      f"shipped decoder's {busy28} issue utilisation (it spends {lz4['valu_per_sequence']}); no design on the "
                      "table lowers the count, the ones that remove the far-match requests raise it. None is projected above "
                      "the shipped decoder; none was built.")
    # ---- what round 6 measured on the card (the two experiments VERDICT r5 asked for, and the design it pointed to) ----
    far = [json.loads(l) for l in open(os.path.join(prof, "r06_far_ablate.jsonl"))]
    tok = json.load(open(os.path.join(prof, "r06_token_index.json")))
    g = {(r["case"], r["lib"]): r.get("GBps") for r in far}
    ci, cn = tok["counters_32768_chunks"]["with_index"], tok["counters_32768_chunks"]["without"]
    out["round6"] = {
        "far_match_loads_folded_onto_resident_lines": {
            "GBps_28_waves": [g[("mix", "libnvcomp")], g[("mix", "farab")]], "GBps_16_waves": [g[("mix", "w4")], g[("mix", "farab_w4")]],
            "reading": "the request rate is not the wall: +4 % with every far-match request served by the L2"},
        "token_index_instead_of_the_chase": {
            "vector_instructions_x": round(ci["SQ_INSTS_VALU"] / cn["SQ_INSTS_VALU"], 3),
            "scalar_instructions_x": round(ci["SQ_INSTS_SALU"] / cn["SQ_INSTS_SALU"], 3),
            "time_x": round(ci["GRBM_GUI_ACTIVE"] / cn["GRBM_GUI_ACTIVE"], 3),
            "waiting_x": round(ci["SQ_WAIT_ANY"] / cn["SQ_WAIT_ANY"], 3),
            "reading": "26 % fewer vector instructions and 21 % MORE time: with 7 waves per SIMD the decoder is bound by its waves' "
                       "dependent chains (LDS and memory round trips in series), which LOOK like vector issue at 0.83-0.88 busy; "
                       "taking instructions out of the chase and putting round trips in (the index's refills, its list) loses"},
        "also_measured_and_not_kept": ["a whole-wave path for runs (sorted-key column 2 166 -> 1 720 GB/s)",
                                       "the chase's leftover tokens kept in a register (mix 648 -> 646, text 596 -> 558)"],
    }
    out["verdict"] = out["verdict"] + (" Round 6: the cycles per instruction are measured (4.1 / 4.17), neither removing the far-match "
                                       "requests (+4 %) nor a quarter of the vector instructions (-21 % throughput) moves the decoder: "
                                       "what is left is the length of a batch's dependent chain at 7 waves per SIMD.")
    json.dump(out, open(os.path.join(prof, "r06_decode_model.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
