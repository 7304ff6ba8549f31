"""Static read of MathNet.Numerics' IL ("missing" 4): KartLQR.cs:104-105 calls LHS.Solve on a sparse-storage matrix; the
oracle restates what that does (oracle/hk_oracle_lq.c: lu_solve) from MathNet's published algorithm.  The binary itself sits in the
reference (Assets/Plugins/MathNet.Numerics.dll): this tool parses its CLI metadata (ECMA-335: PE -> CLI header -> #~ tables -> method
bodies), disassembles the methods on that call path and checks the facts the restatement relies on:

  * Matrix<T>.Solve(Matrix / Vector) factors with LU(); Double.Matrix.LU() is UserLU.Create, and SparseMatrix does not override it;
  * UserLU.Create is the column-oriented (JAMA) Doolittle: s = s + LU[i,k] * col[k] for k < min(i, j), k ascending, multiply THEN add;
    pivot = the first row whose |.| is STRICTLY larger (ble.un skips on <= and on NaN); whole-row swap; the entries below the diagonal
    are DIVIDED by the pivot (no reciprocal);
  * UserLU.Solve(Matrix, Matrix) and (Vector, Vector): row swaps of the right-hand side in pivot order, forward substitution k ascending
    with temp = B[k,j] * LU[i,k]; B[i,j] = B[i,j] - temp, backward substitution k descending with B[k,j] /= LU[k,k] first;
  * (round 4) the sparse products of KartLQR.cs:78-117 — Multiply on two SparseMatrix (DoMultiplySparse, Gustavson's row product),
    TransposeThisAndMultiply with a Matrix (Double.Matrix's dense triple loop: SparseMatrix does not override it) and with a Vector: every
    one accumulates k ASCENDING and as `s = s + a * b`, a multiply and an add with a rounding each (IL `mul add`; Mono does not contract).
    The arithmetic contract of this repository keeps that order and fuses the two roundings (`s = fma(a, b, s)`, which is what
    the fp64 matrix core computes): that one rounding per term is the whole difference to MathNet, bounded in tests/test_lq_numpy_mirror.py.

Nothing of the binary is copied: the tool reads it where it lies, and tests/test_mathnet_il.py keeps only these facts and the file's SHA-256
(tests/golden/mathnet_userlu_facts.json).  usage: python tools/mathnet_il.py [--dump TYPE::METHOD ...] [--update]"""
import hashlib
import json
import os
import re
import struct
import sys

DLL = "/root/reference/Assets/Plugins/MathNet.Numerics.dll"
FACTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mathnet_userlu_facts.json")

N, I1, I4, I8, R4, R8, SW, BR1, BR4, TOK, VAR1, VAR2 = range(12)
OPS, OPS2 = {}, {}


def _reg(start, names, kind=N):
    for k, n in enumerate(names.split()):
        OPS[start + k] = (n, kind)


_reg(0x00, "nop break ldarg.0 ldarg.1 ldarg.2 ldarg.3 ldloc.0 ldloc.1 ldloc.2 ldloc.3 stloc.0 stloc.1 stloc.2 stloc.3")
_reg(0x0E, "ldarg.s ldarga.s starg.s ldloc.s ldloca.s stloc.s", VAR1)
_reg(0x14, "ldnull ldc.i4.m1 ldc.i4.0 ldc.i4.1 ldc.i4.2 ldc.i4.3 ldc.i4.4 ldc.i4.5 ldc.i4.6 ldc.i4.7 ldc.i4.8")
OPS.update({0x1F: ("ldc.i4.s", I1), 0x20: ("ldc.i4", I4), 0x21: ("ldc.i8", I8), 0x22: ("ldc.r4", R4), 0x23: ("ldc.r8", R8), 0x25: ("dup", N),
            0x26: ("pop", N), 0x27: ("jmp", TOK), 0x28: ("call", TOK), 0x29: ("calli", TOK), 0x2A: ("ret", N), 0x45: ("switch", SW)})
_reg(0x2B, "br.s brfalse.s brtrue.s beq.s bge.s bgt.s ble.s blt.s bne.un.s bge.un.s bgt.un.s ble.un.s blt.un.s", BR1)
_reg(0x38, "br brfalse brtrue beq bge bgt ble blt bne.un bge.un bgt.un ble.un blt.un", BR4)
_reg(0x46, "ldind.i1 ldind.u1 ldind.i2 ldind.u2 ldind.i4 ldind.u4 ldind.i8 ldind.i ldind.r4 ldind.r8 ldind.ref stind.ref stind.i1 stind.i2 "
           "stind.i4 stind.i8 stind.r4 stind.r8 add sub mul div div.un rem rem.un and or xor shl shr shr.un neg not conv.i1 conv.i2 conv.i4 "
           "conv.i8 conv.r4 conv.r8 conv.u4 conv.u8")
OPS.update({0x6F: ("callvirt", TOK), 0x70: ("cpobj", TOK), 0x71: ("ldobj", TOK), 0x72: ("ldstr", TOK), 0x73: ("newobj", TOK),
            0x74: ("castclass", TOK), 0x75: ("isinst", TOK), 0x76: ("conv.r.un", N), 0x79: ("unbox", TOK), 0x7A: ("throw", N),
            0x7B: ("ldfld", TOK), 0x7C: ("ldflda", TOK), 0x7D: ("stfld", TOK), 0x7E: ("ldsfld", TOK), 0x7F: ("ldsflda", TOK),
            0x80: ("stsfld", TOK), 0x81: ("stobj", TOK), 0x8C: ("box", TOK), 0x8D: ("newarr", TOK), 0x8E: ("ldlen", N), 0x8F: ("ldelema", TOK),
            0xA3: ("ldelem", TOK), 0xA4: ("stelem", TOK), 0xA5: ("unbox.any", TOK), 0xC2: ("refanyval", TOK), 0xC3: ("ckfinite", N),
            0xC6: ("mkrefany", TOK), 0xD0: ("ldtoken", TOK), 0xDD: ("leave", BR4), 0xDE: ("leave.s", BR1), 0xDF: ("stind.i", N), 0xE0: ("conv.u", N)})
_reg(0x82, "conv.ovf.i1.un conv.ovf.i2.un conv.ovf.i4.un conv.ovf.i8.un conv.ovf.u1.un conv.ovf.u2.un conv.ovf.u4.un conv.ovf.u8.un conv.ovf.i.un conv.ovf.u.un")
_reg(0x90, "ldelem.i1 ldelem.u1 ldelem.i2 ldelem.u2 ldelem.i4 ldelem.u4 ldelem.i8 ldelem.i ldelem.r4 ldelem.r8 ldelem.ref stelem.i stelem.i1 "
           "stelem.i2 stelem.i4 stelem.i8 stelem.r4 stelem.r8 stelem.ref")
_reg(0xB3, "conv.ovf.i1 conv.ovf.u1 conv.ovf.i2 conv.ovf.u2 conv.ovf.i4 conv.ovf.u4 conv.ovf.i8 conv.ovf.u8")
_reg(0xD1, "conv.u2 conv.u1 conv.i conv.ovf.i conv.ovf.u add.ovf add.ovf.un mul.ovf mul.ovf.un sub.ovf sub.ovf.un endfinally")
OPS2.update({0x00: ("arglist", N), 0x01: ("ceq", N), 0x02: ("cgt", N), 0x03: ("cgt.un", N), 0x04: ("clt", N), 0x05: ("clt.un", N),
             0x06: ("ldftn", TOK), 0x07: ("ldvirtftn", TOK), 0x09: ("ldarg", VAR2), 0x0A: ("ldarga", VAR2), 0x0B: ("starg", VAR2),
             0x0C: ("ldloc", VAR2), 0x0D: ("ldloca", VAR2), 0x0E: ("stloc", VAR2), 0x0F: ("localloc", N), 0x11: ("endfilter", N),
             0x12: ("unaligned.", I1), 0x13: ("volatile.", N), 0x14: ("tail.", N), 0x15: ("initobj", TOK), 0x16: ("constrained.", TOK),
             0x17: ("cpblk", N), 0x18: ("initblk", N), 0x1A: ("rethrow", N), 0x1C: ("sizeof", TOK), 0x1D: ("refanytype", N), 0x1E: ("readonly.", N)})


class Assembly:
    """the few metadata tables a method listing needs (ECMA-335 II.22, II.24): TypeRef, TypeDef, MethodDef, MemberRef"""

    def __init__(self, path):
        self.d = d = open(path, "rb").read()
        pe = struct.unpack_from("<I", d, 0x3C)[0]
        assert d[pe:pe + 4] == b"PE\0\0"
        nsec, optsz = struct.unpack_from("<H", d, pe + 6)[0], struct.unpack_from("<H", d, pe + 20)[0]
        opt = pe + 24
        ddir = opt + (96 if struct.unpack_from("<H", d, opt)[0] == 0x10B else 112)
        self.secs = []
        for i in range(nsec):
            so = opt + optsz + 40 * i
            vsz, va, rsz, ro = struct.unpack_from("<IIII", d, so + 8)
            self.secs.append((va, vsz, ro, rsz))
        cli = self.off(struct.unpack_from("<I", d, ddir + 14 * 8)[0])
        md = self.off(struct.unpack_from("<I", d, cli + 8)[0])
        assert d[md:md + 4] == b"BSJB"
        p = md + 16 + struct.unpack_from("<I", d, md + 12)[0]
        nstreams = struct.unpack_from("<H", d, p + 2)[0]
        p += 4
        self.streams = {}
        for _ in range(nstreams):
            o, s = struct.unpack_from("<II", d, p)
            e = d.index(b"\0", p + 8)
            self.streams[d[p + 8:e].decode()] = (md + o, s)
            p = (e + 4) & ~3
        t0 = self.streams["#~"][0]
        heap = d[t0 + 6]
        valid = struct.unpack_from("<Q", d, t0 + 8)[0]
        p = t0 + 24
        self.rows = rows = [0] * 64
        for i in range(64):
            if valid >> i & 1:
                rows[i] = struct.unpack_from("<I", d, p)[0]
                p += 4
        S, G, B = (4 if heap & 1 else 2), (4 if heap & 2 else 2), (4 if heap & 4 else 2)
        idx = lambda t: 4 if rows[t] >= 65536 else 2
        coded = lambda tabs, bits: 4 if max(rows[t] for t in tabs) >= (1 << (16 - bits)) else 2
        TDR, RS, MRP = coded([2, 1, 0x1B], 2), coded([0, 0x1A, 0x23, 1], 2), coded([2, 1, 0x1A, 6, 0x1B], 3)
        self.schema = {0: [2, S, G, G, G], 1: [RS, S, S], 2: [4, S, S, TDR, idx(4), idx(6)], 3: [idx(4)], 4: [2, S, B], 5: [idx(6)],
                       6: [4, 2, 2, S, B, idx(8)], 7: [idx(8)], 8: [2, 2, S], 9: [idx(2), TDR], 10: [MRP, S, B]}
        self.tab_off = {}
        for t in range(11):
            self.tab_off[t] = p
            p += rows[t] * sum(self.schema[t])
        self.types = []
        for i in range(1, rows[2] + 1):
            r = self.row(2, i)
            self.types.append((self.string(r[2]), self.string(r[1]), r[5]))
        self.owner = {}
        for ti in range(len(self.types)):
            for m in self.methods_of(ti):
                self.owner[m] = ti

    def off(self, rva):
        for va, vsz, ro, rsz in self.secs:
            if va <= rva < va + max(vsz, rsz):
                return rva - va + ro
        raise ValueError(hex(rva))

    def rd(self, o, n):
        return int.from_bytes(self.d[o:o + n], "little")

    def row(self, t, i):
        o = self.tab_off[t] + (i - 1) * sum(self.schema[t])
        out = []
        for n in self.schema[t]:
            out.append(self.rd(o, n))
            o += n
        return out

    def string(self, i):
        s0 = self.streams["#Strings"][0]
        return self.d[s0 + i:self.d.index(b"\0", s0 + i)].decode("utf8", "replace")

    def methods_of(self, ti):
        hi = self.types[ti + 1][2] if ti + 1 < len(self.types) else self.rows[6] + 1
        return range(self.types[ti][2], hi)

    def find(self, namespace, type_name, method):
        """MethodDef rows named `method` of the type, in declaration order"""
        for ti, (ns, nm, _) in enumerate(self.types):
            if ns == namespace and nm == type_name:
                return [m for m in self.methods_of(ti) if self.string(self.row(6, m)[3]) == method]
        return []

    def tok_name(self, t):
        tab, i = t >> 24, t & 0xFFFFFF
        if tab == 6:
            ti = self.owner[i]
            return "%s.%s::%s" % (self.types[ti][0].split(".")[-1], self.types[ti][1], self.string(self.row(6, i)[3]))
        if tab == 10:
            r = self.row(10, i)
            tag, ci = r[0] & 7, r[0] >> 3
            owner = self.types[ci - 1][1] if tag == 0 else self.string(self.row(1, ci)[1]) if tag == 1 else "spec"
            return "%s::%s" % (owner, self.string(r[1]))
        if tab == 1:
            return self.string(self.row(1, i)[1])
        if tab == 2:
            return self.types[i - 1][1]
        return "tok"

    def disasm(self, mi):
        d = self.d
        o = self.off(self.row(6, mi)[0])
        if d[o] & 3 == 2:
            size, code = d[o] >> 2, o + 1
        else:
            fl, _, size, _ = struct.unpack_from("<HHII", d, o)
            code = o + (fl >> 12) * 4
        out, p, end = [], code, code + size
        while p < end:
            a, op = p - code, d[p]
            p += 1
            if op == 0xFE:
                name, k = OPS2[d[p]]
                p += 1
            else:
                name, k = OPS[op]
            arg = ""
            if k == I1:
                arg = str(struct.unpack_from("<b", d, p)[0]); p += 1
            elif k == VAR1:
                arg = str(d[p]); p += 1
            elif k == VAR2:
                arg = str(self.rd(p, 2)); p += 2
            elif k == I4:
                arg = str(struct.unpack_from("<i", d, p)[0]); p += 4
            elif k == I8:
                arg = str(struct.unpack_from("<q", d, p)[0]); p += 8
            elif k == R4:
                arg = repr(struct.unpack_from("<f", d, p)[0]); p += 4
            elif k == R8:
                arg = repr(struct.unpack_from("<d", d, p)[0]); p += 8
            elif k == BR1:
                arg = "IL_%04x" % (p + 1 - code + struct.unpack_from("<b", d, p)[0]); p += 1
            elif k == BR4:
                arg = "IL_%04x" % (p + 4 - code + struct.unpack_from("<i", d, p)[0]); p += 4
            elif k == TOK:
                arg = self.tok_name(self.rd(p, 4)); p += 4
            elif k == SW:
                n = self.rd(p, 4)
                p += 4 + 4 * n
            out.append((a, name, arg))
        return out


def shape(listing):
    """the listing as one string of 'opcode operand' items without addresses (branch targets dropped)"""
    return " ".join(n if a.startswith("IL_") or not a else "%s %s" % (n, a) for _, n, a in listing) + " "


AT = r"callvirt spec::At "
V = r"(?:ldloc\.\d|ldloc\.s \d+|ldarg\.\d) "          # a local / argument on the stack
ST = r"(?:stloc\.\d|stloc\.s \d+) "


def check(asm):
    """-> dict of facts (all must hold); raises AssertionError naming the first that does not"""
    F = {}
    NS, FAC = "MathNet.Numerics.LinearAlgebra", "MathNet.Numerics.LinearAlgebra.Double.Factorization"
    # dispatch
    solves = [shape(asm.disasm(m)) for m in asm.find(NS, "Matrix`1", "Solve")]
    F["Matrix<T>.Solve overloads factor with LU() when square"] = sum("::LU " in s for s in solves)
    assert F["Matrix<T>.Solve overloads factor with LU() when square"] >= 2, solves
    lu = asm.find(NS + ".Double", "Matrix", "LU")
    assert len(lu) == 1 and "call Factorization.UserLU::Create " in shape(asm.disasm(lu[0]))
    F["Double.Matrix.LU() calls UserLU.Create"] = True
    assert asm.find(NS + ".Double", "SparseMatrix", "LU") == [] and len(asm.find(NS + ".Double", "DenseMatrix", "LU")) == 1
    F["SparseMatrix does not override LU() (DenseMatrix does)"] = True
    # UserLU.Create
    c = shape(asm.disasm(asm.find(FAC, "UserLU", "Create")[0]))
    assert re.search(r"call Math::Min " + ST + r"ldc\.r8 0\.0 " + ST, c)
    F["Create: kmax = Math.Min(i, j), s = 0.0"] = True
    m = re.search(V + V + V + V + AT + V + V + r"ldelem\.r8 mul add " + ST, c)
    assert m and "ldloc" in m.group(0)
    F["Create: s = s + LU.At(i, k) * col[k]  (mul, then add; k ascending)"] = True
    assert re.search(r"ldelema Double dup ldind\.r8 " + V + r"sub stind\.r8 ", c)
    F["Create: col[i] -= s, then stored to LU[i, j]"] = True
    assert re.search(r"ldelem\.r8 call Math::Abs " + V + V + r"ldelem\.r8 call Math::Abs ble\.un\.s " + V + ST, c)
    F["Create: pivot = first row with STRICTLY larger |col[i]| (ble.un.s skips on <= or NaN)"] = True
    assert len(re.findall(AT, c)) >= 8 and re.search(r"stelem\.i4 ", c)
    assert re.search(AT + r"ldc\.r8 0\.0 ceq ldc\.i4\.0 ceq and brfalse\.s ", c) and re.search(AT + V + V + V + AT + r"div " + AT, c)
    F["Create: if LU[j, j] != 0.0 the entries below are DIVIDED by it"] = True
    assert " mul " in c and c.count(" div ") == 1 and " rem" not in c
    # UserLU.Solve (Matrix, Matrix) and (Vector, Vector)
    for m_row, kind in zip(asm.find(FAC, "UserLU", "Solve"), ("matrix", "vector")):
        s = shape(asm.disasm(m_row))
        assert "::CopyTo " in s and "ldfld spec::Pivots" in s
        body = s[s.index("::CopyTo "):]
        i_swap, i_fwd, i_div = body.index("ldfld spec::Pivots"), body.index(" mul "), body.index(" div ")
        assert i_swap < i_fwd < i_div, kind
        assert body.count(" mul ") == 2 and body.count(" div ") == 1 and body.count(" sub ") >= 3
        if kind == "matrix":      # temp = B[k, j] * LU[i, k]; B[i, j] = B[i, j] - temp
            assert len(re.findall(r"mul " + ST + r".{0,160}?" + V + r"sub callvirt spec::At ", body)) == 2
        else:                     # b[i] = b[i] - b[k] * LU[i, k]
            assert len(re.findall(r"callvirt spec::get_Item " + V + V + r"callvirt spec::get_Item .{0,80}?" + AT + r"mul sub callvirt spec::set_Item ", body)) == 2
        # the forward sweep counts up (add before the mul ... blt), the backward sweep starts at n - 1 and counts down
        assert re.search(r"ldc\.i4\.1 sub " + ST + r"br", body[i_fwd:i_div + 400])
        F["Solve(%s): pivots' row swaps, forward k ascending (temp = B[k]*LU[i,k]; B[i] -= temp), backward k descending with B[k] /= LU[k,k] first" % kind] = True
    # ---- the sparse products of KartLQR.cs:78-117 (every matrix there is CreateMatrix.Sparse: SparseMatrix on CSR storage)
    dm = [shape(asm.disasm(m)) for m in asm.find(NS + ".Double", "SparseMatrix", "DoMultiply")]
    assert sum("call Double.SparseMatrix::DoMultiplySparse " in x for x in dm) == 1
    F["SparseMatrix.DoMultiply(Matrix, Matrix) hands sparse operands to DoMultiplySparse"] = True
    g = shape(asm.disasm(asm.find(NS + ".Double", "SparseMatrix", "DoMultiplySparse")[0]))
    # Gustavson's row product: for row i, for each stored (i, k) of this in storage order (CSR: k ascending), for each stored (k, j) of other:
    # first touch of column j in this row: c[j] = a * b (stored as it is); later: c[j] = c[j] + a * b  (mul, then add: two roundings)
    assert re.search(V + V + r"mul stelem\.r8 ", g) and re.search(r"ldelema Double dup ldind\.r8 " + V + V + r"mul add stind\.r8 ", g)
    assert g.count(" mul ") == 2 and " sub " not in g and " div " not in g and "callvirt spec::Normalize" in g
    assert len(re.findall(r"ldc\.i4\.1 add " + ST, g)) >= 6 and "ldc.i4.1 sub" not in g          # every loop counts up
    F["DoMultiplySparse: rows of this, its stored entries k ascending, other's row k; c[j] = a*b on first touch, else c[j] = c[j] + a*b (mul, then add)"] = True
    tt = asm.find(NS + ".Double", "SparseMatrix", "DoTransposeThisAndMultiply")
    assert len(tt) == 1
    t = shape(asm.disasm(tt[0]))
    assert "callvirt spec::get_Item" in t and re.search(r"callvirt spec::get_Item " + V + V + r"ldelem\.r8 " + V + r"mul add callvirt spec::set_Item ", t) and "ldc.i4.1 sub" not in t
    F["SparseMatrix.DoTransposeThisAndMultiply exists for (Vector, Vector) only: rows k ascending, result[j] = result[j] + value * right[k] (mul, then add)"] = True
    base = [shape(asm.disasm(m)) for m in asm.find(NS + ".Double", "Matrix", "DoTransposeThisAndMultiply")]
    mm = [b for b in base if b.count("spec::At ") == 3]
    assert len(mm) == 1 and re.search(r"ldc\.r8 0\.0 " + ST, mm[0]) and re.search(V + V + V + V + r"call spec::At " + V + V + V + r"callvirt spec::At mul add " + ST, mm[0])
    assert "ldc.i4.1 sub" not in mm[0] and mm[0].count(" mul ") == 1
    F["TransposeThisAndMultiply(Matrix) on a SparseMatrix runs Double.Matrix's loop: s = 0.0; s = s + this.At(k, j) * other.At(k, i) for EVERY k ascending (mul, then add)"] = True
    return F


def main(argv):
    if not os.path.exists(DLL):
        raise SystemExit("the reference's MathNet.Numerics.dll is not here: nothing to read")
    asm = Assembly(DLL)
    if "--dump" in argv:
        for spec in argv[argv.index("--dump") + 1:]:
            tname, mname = spec.split("::")
            ns, _, tn = tname.rpartition(".")
            for m in asm.find(ns, tn, mname):
                print("==== %s (MethodDef row %d)" % (spec, m))
                for a, n, arg in asm.disasm(m):
                    print("IL_%04x: %s %s" % (a, n, arg))
        return
    facts = check(asm)
    out = {"file": "Assets/Plugins/MathNet.Numerics.dll", "sha256": hashlib.sha256(asm.d).hexdigest(), "facts": facts,
           "restated_in": "oracle/hk_oracle_lq.c: lu_solve; csrc/hk_lq_core.h (same order)"}
    for k, v in facts.items():
        print("ok  %s%s" % (k, "" if v is True else " (%s)" % v))
    if "--update" in argv:
        with open(FACTS, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
