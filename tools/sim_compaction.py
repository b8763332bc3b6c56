#!/usr/bin/env python3
"""Back-of-the-envelope simulation of wave-level work compaction for trt_trace on incoherent rays (regime 2).

The kernel is VALU issue-bound (profiles/r02_pmc_trace_aimed.json), so its time is the number of wave-instructions
executed, and a wave-instruction costs the same whether 6 or 64 lanes are active.  This script counts wave-instructions
per 64 rays for three organisations, with per-stage instruction counts read off the ISA (`make asm`) and a ray population
like tools/bench_configs.py's aimed set (30 % culled in setup, 45 % hit, the rest pass the bounding box and miss):

  static        today's kernel: one ray per lane, every stage runs for the whole wave, the walk runs to its slowest lane
  pool(S)       a pool of S ray slots per wave (S = 64: registers; more: LDS), stages executed 64 lanes at a time on the
                stage with most waiting rays, `move` = extra instructions per stage execution for getting state in and out
  queues(K)     K rays per lane, lane-private refill at trip granularity, two trip kinds (Newton / service), the service
                trip runs when at least `thresh` lanes wait for one

Result (printed): with 64 slots nothing is gained; 128-256 slots give 1.45-1.6x before and 1.15-1.25x after a realistic
`move` cost; per-lane queues of 4-8 rays give 1.1-1.3x on the walk part only.  None of them was built."""
import random

COST = dict(load=10, setup=110, setup_cull=60, enter=40, newton=24, finish=35, out_hit=66, out_miss=16)
STAGES = ["load", "setup", "enter", "newton", "finish", "out"]


def make_ray(rng):
    if rng.random() < 0.30:
        return ["load", "setup", "out"], False
    hit = rng.random() < 0.64
    seq = ["load", "setup"]
    if hit:
        pieces = 1 if rng.random() < 0.7 else 2
        for p in range(pieces):
            seq.append("enter")
            seq += ["newton"] * (rng.choice([2, 3, 3, 4, 4, 5]) if p == pieces - 1 else rng.choice([1, 1, 2]))
        seq += ["finish", "out"]
    else:
        for _ in range(rng.choice([1, 2, 2, 3])):
            seq.append("enter")
            seq += ["newton"] * rng.choice([0, 1, 1, 2])
        seq += ["out"]
    return seq, hit


def static(n, rng):
    total = 0
    for _ in range(n // 64):
        rays = [make_ray(rng) for _ in range(64)]
        t = COST["load"] + COST["setup"]
        t += max(sum(1 for s in r[0] if s in ("enter", "newton")) for r in rays) * 45
        t += COST["finish"] + COST["out_hit"] if any(r[1] for r in rays) else COST["out_miss"]
        total += t
    return total / (n // 64)


def pool(n, slots, move, rng):
    rays = [make_ray(rng) for _ in range(n)]
    nxt, lane, total = 0, [None] * slots, 0
    while True:
        groups = {s: [] for s in STAGES}
        for i, l in enumerate(lane):
            if l is None:
                if nxt < n:
                    groups["load"].append(i)
            else:
                groups[l[0][l[1]]].append(i)
        if not any(groups.values()):
            break
        s = max(STAGES, key=lambda k: (min(64, len(groups[k])), STAGES.index(k)))
        idx = groups[s][:64]
        if s == "load":
            for i in idx[:max(0, n - nxt)]:
                lane[i] = [rays[nxt][0], 1, rays[nxt][1]]
                nxt += 1
            total += COST["load"] + move
            continue
        total += (COST["setup"] if s == "setup" else COST["out_hit"] if s == "out" else COST[s]) + move
        for i in idx:
            lane[i][1] += 1
            if lane[i][1] >= len(lane[i][0]):
                lane[i] = None
    return total / (n / 64)


def queues(K, thresh, rng, chunks=300, c_newton=34, c_service=62):
    """walk part only: per-lane queues of K set-up rays, Newton trips vs service trips"""
    def ray():
        if rng.random() < 0.30:
            return None
        if rng.random() < 0.64:
            np_ = 1 if rng.random() < 0.7 else 2
            return [rng.choice([1, 2, 2, 3, 3, 4]) if p == np_ - 1 else rng.choice([0, 1, 1]) for p in range(np_)]
        return [rng.choice([0, 0, 1, 1, 2]) for _ in range(rng.choice([1, 2, 2, 3]))]
    total = nrays = 0
    for _ in range(chunks):
        st = []
        for _ in range(64):
            q = [r for r in (ray() for _ in range(K)) if r is not None]
            st.append({"q": q, "cur": 0, "piece": 0, "rem": 0, "phase": "S" if q else "I"})
        nrays += 64 * K
        while True:
            n_s = sum(1 for s in st if s["phase"] == "S")
            n_n = sum(1 for s in st if s["phase"] == "N")
            if not n_s and not n_n:
                break
            if n_n == 0 or n_s >= thresh:
                total += c_service
                for s in st:
                    if s["phase"] != "S":
                        continue
                    while True:
                        if s["cur"] >= len(s["q"]):
                            s["phase"] = "I"
                            break
                        r = s["q"][s["cur"]]
                        if s["piece"] >= len(r):
                            s["cur"] += 1
                            s["piece"] = 0
                            continue
                        s["rem"] = r[s["piece"]]
                        s["piece"] += 1
                        s["phase"] = "N" if s["rem"] > 0 else "S"
                        break
            else:
                total += c_newton
                for s in st:
                    if s["phase"] == "N":
                        s["rem"] -= 1
                        if s["rem"] == 0:
                            s["phase"] = "S"
    return total / (nrays / 64)


if __name__ == "__main__":
    rng = random.Random(1)
    print(f"static (today's organisation)              {static(64 * 400, rng):6.0f} wave-instructions per 64 rays")
    for slots in (64, 128, 192, 256):
        print(f"pool of {slots:3d} slots, move cost 0 / 10 / 20   " + " / ".join(f"{pool(64 * 300, slots, m, rng):4.0f}" for m in (0, 10, 20)))
    print("walk part only (today: max over the wave x 45 = ~350-400):")
    for K in (2, 4, 8):
        print(f"per-lane queues of {K} rays, service threshold 16 / 32 / 48   " + " / ".join(f"{queues(K, t, rng):4.0f}" for t in (16, 32, 48)))
