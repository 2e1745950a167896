"""Test-side reader of the reference's textual command IR, written independently of the product's C++ parser
(divans_amd/csrc/ir.cpp) from the grammar in src/bin/divans.rs:191-483, so the two check each other.  Test infrastructure."""
import lzma
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = {"lsb6": 0, "msb6": 1, "utf8": 2, "sign": 3}


def load_ir_text(name):
    with lzma.open(os.path.join(GOLDEN, f"ir_{name}.ir.xz")) as f:
        return f.read()


def parse(text):
    """-> list of commands: ("insert", bytes) ("copy", n, dist) ("dict", bytes) ("ltype", bt, stride) ("ctype", bt) ("dtype", bt) ("prediction", dict)"""
    cmds = []
    for line in text.decode("latin-1").split("\n"):
        f = line.split(" ")
        op = f[0]
        if op in ("", "window"):
            continue
        if op == "insert":
            if int(f[1]) == 0:
                continue
            data = bytes.fromhex(f[2]); assert len(data) == int(f[1])
            cmds.append(("insert", data))
        elif op == "copy":
            assert f[2] == "from"
            if int(f[1]):
                cmds.append(("copy", int(f[1]), int(f[3])))
        elif op == "dict":
            k = f.index("func")
            data = bytes.fromhex(f[k + 2]); assert len(data) == int(f[1]) & 0xff
            cmds.append(("dict", data))
        elif op == "ltype":
            cmds.append(("ltype", int(f[1]) & 0xff, int(f[2]) if len(f) > 2 else 0))
        elif op in ("ctype", "dtype"):
            cmds.append((op, int(f[1]) & 0xff))
        elif op == "prediction":
            def nums(key):
                if key not in f:
                    return []
                out = []
                for x in f[f.index(key) + 1:]:
                    if not x.isdigit():
                        break
                    out.append(int(x))
                return out
            cmds.append(("prediction", dict(mode=MODES[f[1]], lcontextmap=nums("lcontextmap"), dcontextmap=nums("dcontextmap"),
                                            mixingvalues=nums("mixingvalues"),
                                            speeds={k: nums(k)[:2] for k in ("cmspeedinc", "cmspeedmax", "stspeedinc", "stspeedmax", "mxspeedinc", "mxspeedmax")})))
        else:
            raise ValueError("unknown IR line: " + line[:50])
    return cmds


def expand(cmds):
    """(raw bytes, literal bytes, segments [(len, btype, last8)])"""
    out = bytearray(); lit = bytearray(); segs = []; bt = 0
    for c in cmds:
        if c[0] == "insert":
            tail = bytes(out[-8:]).rjust(8, b"\0")
            segs.append((len(c[1]), bt, int.from_bytes(tail, "little")))
            out += c[1]; lit += c[1]
        elif c[0] == "copy":
            for _ in range(c[1]):
                out.append(out[-c[2]])
        elif c[0] == "dict":
            out += c[1]
        elif c[0] == "ltype":
            bt = c[1]
    return (np.frombuffer(bytes(out), dtype=np.uint8), np.frombuffer(bytes(lit), dtype=np.uint8), segs)
