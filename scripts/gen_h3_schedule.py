"""Generates hold_amd/csrc/rmlp_h3_sched.h: the order in which the epilogue micro-operations of csrc/rmlp_h3.hip are placed
behind the 24 MFMAs of a k step.

Why a table.  One wave per SIMD issues in order, and on this chip a lone wave issues roughly one instruction of ANY kind per
7-8 cycles (measured, DESIGN.md section 4.1c: cycles per k step / instructions per k step is 6.8-7.3 for rmlp.hip, rmlp_h3.hip
and wgrad_r6.hip alike), so a k step costs what its instruction COUNT costs and the matrix pipe (24 MFMAs x 32 cycles) is busy
only as far as the other ~100 instructions of the step are spread evenly between the MFMAs: a gap that receives a whole round
of transcendentals stalls everything behind it.  The micro-operations keep the indices of rmlp_h3.hip (round-major: the
softplus rounds, then the limb split, then the stores); this script list-schedules them into the 24 gaps under a budget of
instruction slots per gap (fixed slots: the MFMA, the fragment read of gaps 0..3 of a group, the LDS wait at the head of a
group, the DMA pair of gap 1 of groups 2 and 3, the rendezvous before group 2), keeps a dependent operation at least LAT
slots behind its producer (transcendentals: LAT_TRANS -- their results are consumed by plain VALU code and the assembler-
level v_exp / v_log are opaque to the compiler's hazard recogniser) and at most one transcendental per gap, and prints the
permutation and the gap boundaries.

  python scripts/gen_h3_schedule.py > hold_amd/csrc/rmlp_h3_sched.h
"""
NGAP = 24
LAT_PLAIN, LAT_TRANS = 2, 4  # instruction slots between a producer and its first consumer


def rounds(head):
    """[(name, n, cost, trans, deps)]: deps(i) -> [(round, idx)] of round-local operation i.  Values 0..7, pairs 0..3."""
    pair = lambda r: (lambda p: [(r, 2 * p), (r, 2 * p + 1)])
    if head:
        R = [("acc", 8, 1, False, lambda i: []),
             ("ys", 4, 1, False, pair(0)),
             ("earg", 8, 1, False, lambda i: [(1, i >> 1)]),
             ("exp", 8, 2, True, lambda i: [(2, i)]),
             ("add", 4, 1, False, pair(3)),
             ("log", 8, 2, True, lambda i: [(4, i >> 1)]),
             ("relu", 8, 1, False, lambda i: [(1, i >> 1)]),
             ("fma", 4, 1, False, lambda p: [(5, 2 * p), (5, 2 * p + 1), (6, 2 * p), (6, 2 * p + 1)])]
        last = 7
    else:
        R = [("acc", 8, 1, False, lambda i: []),
             ("ys", 4, 1, False, pair(0)),
             ("earg", 8, 1, False, lambda i: [(1, i >> 1)]),
             ("exp", 8, 2, True, lambda i: [(2, i)]),
             ("add", 4, 1, False, pair(3)),
             ("ser1", 4, 1, False, pair(3)),
             ("ser2", 4, 1, False, lambda p: [(5, p)]),
             ("e001", 4, 1, False, pair(3)),
             ("ser3", 4, 1, False, lambda p: [(6, p), (7, p)]),
             ("log", 8, 2, True, lambda i: [(4, i >> 1)]),
             ("lgs", 4, 1, False, pair(9)),
             ("sel", 8, 2, False, lambda i: [(10, i >> 1), (8, i >> 1)]),
             ("relu", 8, 1, False, lambda i: [(1, i >> 1)]),
             ("sum", 4, 1, False, lambda p: [(11, 2 * p), (11, 2 * p + 1), (12, 2 * p), (12, 2 * p + 1)]),
             ("thr", 8, 0, False, lambda i: [(13, i >> 1)]),  # skip-layer override only (the reference's y > 0.2 branch is a
             # no-op here: softplus(y) - y <= 2.1e-11 is far below half an ulp of y >= 0.2, the sum rounds to y)
             ("out", 4, 1, False, pair(14))]
        last = 14
    return R, last


def build(head):
    R, last = rounds(head)
    base = [0]
    for _, n, _, _, _ in R:
        base.append(base[-1] + n)
    ops = {}
    for r, (name, n, cost, trans, deps) in enumerate(R):
        for i in range(n):
            ops[base[r] + i] = dict(cost=cost, trans=trans, deps=[base[rr] + ii for rr, ii in deps(i)], name=name)
    b = base[-1]
    val = lambda i: base[last] + (i if R[last][1] == 8 else i >> 1)  # the operation that finishes value i
    for d in range(4):
        s = lambda op: b + 4 * op + d
        # hi: v_cvt_pk_f16_f32 + the v_max3_f32 that keeps the exact maximum of the scaled activations (overflow guard, round 6)
        ops[s(0)] = dict(cost=2, trans=False, deps=[val(2 * d), val(2 * d + 1)], name="hi")
        ops[s(1)] = dict(cost=1, trans=False, deps=[s(0)], name="ra")
        ops[s(2)] = dict(cost=1, trans=False, deps=[s(0)], name="rb")
        ops[s(3)] = dict(cost=1, trans=False, deps=[s(1), s(2)], name="lo")
    if not head:
        for h2 in range(2):
            ops[b + 16 + h2] = dict(cost=2, trans=False, deps=[base[15] + 2 * h2, base[15] + 2 * h2 + 1], name="store")
    return ops, base


def fixed_slots():
    f = []
    for G in range(NGAP):
        pair, m = divmod(G, 6)
        c = 1  # the MFMA
        if m < 4:
            c += 1  # ds_read_b128 of the next group's fragments
        if m == 0:
            c += 1  # s_waitcnt lgkmcnt(0): the fragments of this group
        if pair >= 2 and m == 1:
            c += 7  # DMA pair: 3 scalar address instructions, s_mov m0, s_nop, two global_load_lds
        if pair == 2 and m == 0:
            c += 2  # rendezvous: s_waitcnt vmcnt + s_barrier
        f.append(c)
    return f


def schedule(head, budget):
    ops, _ = build(head)
    users = {k: [] for k in ops}
    for k, o in ops.items():
        for d in o["deps"]:
            users[d].append(k)
    prio = {}

    def cp(k):
        if k not in prio:
            prio[k] = ops[k]["cost"] + (LAT_TRANS if ops[k]["trans"] else LAT_PLAIN) + max([cp(u) for u in users[k]], default=0)
        return prio[k]

    for k in ops:
        cp(k)
    fixed = fixed_slots()
    done_at, order, gap_end = {}, [], []
    t = 0
    remaining = set(ops)
    for G in range(NGAP):
        t0 = t
        t += fixed[G]
        gaps_left = NGAP - G
        placed_t = False
        while remaining:
            rem = sum(ops[k]["cost"] for k in remaining) + sum(fixed[G + 1:])
            target = max(budget, (rem + (t - t0)) / gaps_left)
            ready = [k for k in remaining if all(d in done_at and done_at[d] <= t for d in ops[k]["deps"])]
            if not ready:
                if G < NGAP - 1:
                    break
                nxt = [k for k in remaining if all(d in done_at for d in ops[k]["deps"])]
                t = min(max(done_at[d] for d in ops[k]["deps"]) for k in nxt)
                continue
            cand = [k for k in ready if not (ops[k]["trans"] and placed_t)] or ready
            k = max(cand, key=lambda k: (prio[k], -k))
            if (t - t0) + ops[k]["cost"] > target and G < NGAP - 1:
                break
            order.append(k)
            remaining.discard(k)
            placed_t = placed_t or ops[k]["trans"]
            t += ops[k]["cost"]
            done_at[k] = t + (LAT_TRANS if ops[k]["trans"] else LAT_PLAIN)
        gap_end.append(len(order))
    assert not remaining
    pos = {k: i for i, k in enumerate(order)}
    for k, o in ops.items():
        assert all(pos[d] < pos[k] for d in o["deps"])
    return order, gap_end, ops, fixed


def emit(name, order, gap_end):
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
        for budget in [b / 2 for b in range(4, 40)]:
            try:
                order, gap_end, ops, fixed = schedule(head, budget)
            except AssertionError:
                continue
            prev, loads = 0, []
            for G, e in enumerate(gap_end):
                loads.append(fixed[G] + sum(ops[k]["cost"] for k in order[prev:e]))
                prev = e
            key = (sum(l * l for l in loads), max(loads))  # the most even spread
            if best is None or key < best[0]:
                best = (key, budget, order, gap_end, loads)
        key, budget, order, gap_end, loads = best
        print(f"// {name}: {len(order)} micro-operations, budget {budget}; instruction slots per gap (fixed ones included): {loads}")
        emit(name, order, gap_end)


if __name__ == "__main__":
    main()
