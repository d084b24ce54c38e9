"""Generates hold_amd/csrc/rmlp_h3_sched.h: the order in which the epilogue micro-operations of csrc/rmlp_h3.hip are placed
behind the 24 MFMAs of a k step.

Why a table: one wave per SIMD issues in order, an MFMA occupies the matrix pipe for 32 cycles, and the wave must be back at
the next MFMA by then.  With half the MFMAs of rmlp.hip per k step the epilogue of the next step's 8 values (softplus + limb
split: 64 plain VALU instructions of 4 cycles and 16 transcendentals of 16 cycles in the sampler query) fills 26 of those 32
cycles ON AVERAGE, so it has to be spread evenly: a gap that receives a whole round of four v_exp_f32 (64 cycles) drains the
pipe.  The micro-operations keep the indices of rmlp_h3.hip (softplus part: 8 round + value; limb split: 8 NR + 4 op +
dword; stores behind); this script list-schedules them into the 24 gaps under a cycle budget per gap (fixed costs: the
fragment read of gaps 0..3 of a group, the DMA pieces of gaps 1 and 4 of groups 2 and 3, the rendezvous before group 2)
with dependent operations at least `LAT` cycles apart, and prints the permutation and the gap boundaries.

  python scripts/gen_h3_schedule.py > hold_amd/csrc/rmlp_h3_sched.h
"""
import sys

PLAIN, TRANS, STORE_C = 4, 16, 12
LAT_PLAIN, LAT_TRANS = 8, 24   # issue-to-use distance (cycles) the scheduler keeps between dependent operations
NGAP = 24


def build(head):
    """-> ops: list of dict(idx, cost, deps, trans)"""
    NR = 8 if head else 16
    ops = {}

    def add(idx, cost, deps, trans=False):
        ops[idx] = dict(idx=idx, cost=cost, deps=list(deps), trans=trans)

    for i in range(8):
        r = lambda rd: 8 * rd + i
        add(r(0), PLAIN, [])            # accvgpr read
        add(r(1), PLAIN, [r(0)])        # ys = c3 y
        add(r(2), PLAIN, [r(1)])        # KE |ys|
        add(r(3), TRANS, [r(2)], True)  # exp2
        add(r(4), PLAIN, [r(3)])        # 1 + e
        if head:
            add(r(5), TRANS, [r(4)], True)  # log2
            add(r(6), PLAIN, [r(1)])        # max(ys, 0)
            add(r(7), PLAIN + (2 if False else 0), [r(5), r(6)])  # CL log2 + max (+ skip override)
            last = r(7)
        else:
            add(r(5), PLAIN, [r(3)])            # series
            add(r(6), PLAIN, [r(5)])
            add(r(7), PLAIN, [r(3)])            # 0.01 SA e
            add(r(8), PLAIN, [r(6), r(7)])
            add(r(9), TRANS, [r(4)], True)      # log2
            add(r(10), PLAIN, [r(9)])           # CL log2
            add(r(11), 2 * PLAIN, [r(10), r(8)])  # select (v_cmp + v_cndmask)
            add(r(12), PLAIN, [r(1)])           # max(ys, 0)
            add(r(13), PLAIN, [r(12), r(11)])
            add(r(14), 2 * PLAIN, [r(13)])      # threshold select (+ skip override)
            add(r(15), PLAIN, [r(14)])          # 1 / SA: the stored value
            last = r(14)
        ops[last]["is_r"] = i
    base = 8 * NR
    rlast = {ops[k]["is_r"]: k for k in ops if "is_r" in ops[k]}
    for d in range(4):
        s = lambda op: base + 4 * op + d
        add(s(0), PLAIN, [rlast[2 * d], rlast[2 * d + 1]])  # cvt_pk hi
        add(s(1), PLAIN, [s(0)])                            # x0 - hi
        add(s(2), PLAIN, [s(0)])                            # x1 - hi
        add(s(3), PLAIN, [s(1), s(2)])                      # cvt_pk lo
    if not head:
        for h2 in range(2):
            add(base + 16 + h2, STORE_C, [8 * 15 + 4 * h2 + i for i in range(4)])
    return ops


def schedule(head, budget, stagger=0):
    ops = build(head)
    n = len(ops)
    # critical-path priority
    users = {k: [] for k in ops}
    for k, o in ops.items():
        for d in o["deps"]:
            users[d].append(k)
    prio = {}

    def cp(k):
        if k not in prio:
            prio[k] = ops[k]["cost"] + max([cp(u) for u in users[k]], default=0)
        return prio[k]

    for k in ops:
        cp(k)
    NR = 8 if head else 16

    def vof(k):  # the value (0..7) an operation belongs to: staggering the values' chains spreads the transcendentals
        if k < 8 * NR:
            return k % 8
        if k < 8 * NR + 16:
            return 2 * ((k - 8 * NR) % 4)
        return 4 * (k - 8 * NR - 16)

    fixed = []
    for G in range(NGAP):
        pair, m = divmod(G, 6)
        c = 4  # the MFMA's own issue
        if m < 4:
            c += 4  # ds_read_b128 of the next group's fragments
        if pair >= 2 and m in (1, 4):
            c += 6  # DMA piece (s_mov m0, s_nop, global_load_lds + address arithmetic)
        if pair == 2 and m == 0:
            c += 8  # rendezvous
        fixed.append(c)
    done_at = {}  # op -> cycle its result is usable
    order, gap_end = [], []
    t = 0
    remaining = set(ops)
    for G in range(NGAP):
        t0 = max(t, G * 32)  # a gap starts when its MFMA can issue: the pipe is free and the previous gap's work is done
        t = t0 + fixed[G]
        gaps_left = NGAP - G
        while remaining:
            # work that must still be placed per remaining gap decides how full this gap gets
            rem_cost = sum(ops[k]["cost"] for k in remaining)
            target = max(budget, rem_cost / gaps_left + fixed[G]) if G < NGAP - 1 else 1e9
            ready = [k for k in remaining if all(d in done_at and done_at[d] <= t for d in ops[k]["deps"])]
            if not ready:
                nxt = [k for k in remaining if all(d in done_at for d in ops[k]["deps"])]
                if not nxt or G < NGAP - 1:
                    break
                k = min(nxt, key=lambda k: max(done_at[d] for d in ops[k]["deps"]))
                t = max(done_at[d] for d in ops[k]["deps"])
                ready = [k]
            # at most one transcendental per gap unless nothing else is ready
            have_t = any(ops[k]["trans"] for k in order[gap_end[-1] if gap_end else 0:])
            cand = [k for k in ready if not (ops[k]["trans"] and have_t)] or ready
            k = max(cand, key=lambda k: (prio[k] + stagger * (7 - vof(k)), -k))
            if (t - t0) + ops[k]["cost"] > target and G < NGAP - 1:
                break
            order.append(k)
            remaining.discard(k)
            t += ops[k]["cost"]
            done_at[k] = t + (LAT_TRANS if ops[k]["trans"] else LAT_PLAIN) - ops[k]["cost"]
        gap_end.append(len(order))
    assert not remaining and len(order) == n, (len(order), n)
    # verify: dependencies precede users
    pos = {k: i for i, k in enumerate(order)}
    for k, o in ops.items():
        assert all(pos[d] < pos[k] for d in o["deps"])
    return order, gap_end, ops, fixed


def emit(name, order, gap_end):
    print(f"// {name}: {len(order)} micro-operations")
    print(f"#define {name}_N {len(order)}")
    print(f"#define {name}_ORDER {{{', '.join(map(str, order))}}}")
    print(f"#define {name}_END {{{', '.join(map(str, gap_end))}}}")


def main():
    print("// GENERATED by scripts/gen_h3_schedule.py -- do not edit.  Placement of the epilogue micro-operations of rmlp_h3.hip")
    print("// behind the 24 MFMAs of a k step: ORDER = the micro-operation indices in issue order, END[G] = number of")
    print("// micro-operations issued up to and including gap G.")
    print("#pragma once")
    for head, name in ((True, "H3_SCHED_HEAD"), (False, "H3_SCHED_STORE")):
        best = None
        for budget in range(16, 64, 2):
            for stagger in (0, 2, 4, 6, 8, 10, 12, 16, 20, 24, 32):
                try:
                    order, gap_end, ops, fixed = schedule(head, budget, stagger)
                except AssertionError:
                    continue
                prev, worst = 0, 0
                for G, e in enumerate(gap_end):
                    worst = max(worst, fixed[G] + sum(ops[k]["cost"] for k in order[prev:e]))
                    prev = e
                if best is None or worst < best[0]:
                    best = (worst, budget, stagger, order, gap_end, ops, fixed)
        worst, budget, stagger, order, gap_end, ops, fixed = best
        print(f"// {name}: budget {budget}, stagger {stagger}, fullest gap {worst} cycles (fixed costs included)")
        emit(name, order, gap_end)
        prev = 0
        cyc = []
        for e in gap_end:
            cyc.append(sum(ops[k]["cost"] for k in order[prev:e]))
            prev = e
        print(f"// cycles of micro-operations per gap: {cyc}  (total {sum(cyc)})", file=sys.stdout)


if __name__ == "__main__":
    main()
