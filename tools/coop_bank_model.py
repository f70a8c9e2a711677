"""Static model of the LDS bank conflicts of the cooperative interpreter (csrc/elp/coop.h): walks the scheduled programs of tools/gen_coop.py and counts,
for every register-file access of every step, the LDS cycles of each 32-lane half (`ds_read_b32` / `ds_write_b32`: bank = word address mod 32, lanes of one half
conflict, equal addresses broadcast -- /opt/skills/guides/cdna_hip_programming.md section 2).  Prints ideal cycles (one per access and half), modelled cycles, and the
share that is conflicts, for the register-file layout in use and for alternatives.  Usage: python tools/coop_bank_model.py [bn254|bls12_381]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_coop as gc  # noqa: E402


def word_interleaved(nl, nreg):
    return lambda reg, comp: (reg * 2 + comp) * nl


def word_planar(nl, nreg):
    plane = nreg * nl
    plane += (16 - plane) % 32           # plane offset = 16 (mod 32): the two lanes of a pair write banks 16 apart
    return lambda reg, comp: comp * plane + reg * nl


def cycles(addrs):
    """LDS cycles of one access of one half: addrs = word addresses of the active lanes (limb 0; the other limbs shift every bank alike)."""
    if not addrs:
        return 0, 0
    per_bank = collections.defaultdict(set)
    for a in addrs:
        per_bank[a % 32].add(a)
    return 1, max(len(s) for s in per_bank.values())


def model(ops, steps, reg, np_, word, nl, sqr_ok):
    ideal = total = 0
    halves = max(1, np_ // 16)          # lane pairs of one item per 32-lane half: 16
    for cls, lst in steps:
        sq_step = cls == 1 and all(i in sqr_ok for i in lst)
        for h in range(halves):
            slots = lst[h * 16:(h + 1) * 16] if np_ > 16 else lst
            if not slots:
                continue
            accesses = []                # each: list of addresses over the active lanes of the half
            if cls == 1:
                a0, a1, y, w, st = [], [], [], [], []
                for i in slots:
                    op, a, b, aux = ops[i]
                    for comp in (0, 1):
                        a0.append(word(reg[a], 0))
                        a1.append(word(reg[a], 1))
                        st.append(word(reg[i], comp))
                        if sq_step:
                            continue
                        if op == gc.MUL:
                            y.append(word(reg[b], comp))
                            w.append(word(reg[b], comp ^ 1))
                        elif op == gc.MULS:
                            so, zo = word(reg[b], aux), word(gc.IN_ONE, 1)
                            y.append(zo if comp else so)
                            w.append(so if comp else zo)
                        # MULC: constants live in another LDS array, every lane its own constant: modelled as conflict-free
                accesses = [a0, a1, st] + ([] if sq_step else [y, w])
            else:
                lists = []
                for i in slots:
                    op, a, b, aux = ops[i]
                    if op == gc.LIN:
                        e0, e1 = gc.lin_entries(aux, reg, nl)
                        # entries carry the interleaved word offset; map back to (reg, comp)
                        conv = lambda e: [word((o // nl) // 2, (o // nl) % 2) for o, _ in e]
                        lists.append(conv(e0))
                        lists.append(conv(e1))
                        accesses.append(None)
                    elif op == gc.INV:
                        lists.append([word(reg[a], 0)])
                        lists.append([])
                    else:                # LDL: from the staged lines
                        lists.append([])
                        lists.append([])
                depth = max((len(l) for l in lists), default=0)
                accesses = [[l[t] for l in lists if t < len(l)] for t in range(depth)]
                accesses.append([word(reg[i], comp) for i in slots for comp in (0, 1)])
            for acc in accesses:
                if not acc:
                    continue
                i_, c_ = cycles(acc)
                ideal += i_ * nl
                total += c_ * nl
    return ideal, total


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    cv = gc.BN254 if name == "bn254" else gc.BLS12_381
    sys.setrecursionlimit(100000)
    res = gc.validate(cv)
    nl = gc.LIMBS[name][0]
    for pname, (prog, steps, reg, outs_c, peak, nmul, nlin, np_) in res.items():
        sqr_ok = {i for i, o in enumerate(prog.ops) if gc.is_square(o) and prog.vb[o[1]] <= gc.MAGS[name][0] / 2}
        for lname, mk in (("interleaved R[reg][comp][limb]", word_interleaved), ("planar R[comp][reg][limb]", word_planar)):
            ideal, total = model(prog.ops, steps, reg, np_, mk(nl, gc.NREG), nl, sqr_ok)
            print("%-8s %-32s ideal %7d  modelled %7d LDS cycles per item-half  conflicts %4.1f %% of the cycles" % (pname, lname, ideal, total, 100.0 * (total - ideal) / total))


if __name__ == "__main__":
    main()
