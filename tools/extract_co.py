"""Pull the gfx950 code object out of a hipcc-built shared library (the bundle in .hip_fatbin) and print the
per-kernel register / spill / scratch notes: python tools/extract_co.py <lib.so> [out.co] [name filter]"""
import struct, subprocess, sys

def extract(lib, out):
    b = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = b.find(magic)
    assert i >= 0, "no offload bundle"
    n = struct.unpack_from("<Q", b, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p)
        triple = b[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple:
            open(out, "wb").write(b[i + off:i + off + size])
            return out
    raise SystemExit("no gfx950 entry")

if __name__ == "__main__":
    lib = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/engine.co"
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    extract(lib, out)
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    cur = {}
    keys = (".name", ".sgpr_count", ".sgpr_spill_count", ".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size")
    rows = []
    for line in txt.splitlines():
        t = line.strip()
        if t.startswith("- .") or t.startswith("."):
            t = t.lstrip("- ")
        for k in keys:
            if t.startswith(k + ":"):
                cur[k] = t.split(":", 1)[1].strip()
        if t.startswith(".wavefront_size"):
            if cur.get(".name") and flt in cur[".name"]:
                rows.append(dict(cur))
            cur = {}
    for r in rows:
        print(f"{r.get('.name','?')[:70]:70s} sgpr {r.get('.sgpr_count','?'):>4} (spill {r.get('.sgpr_spill_count','?'):>3})  vgpr {r.get('.vgpr_count','?'):>4} (spill {r.get('.vgpr_spill_count','?'):>3})  scratch {r.get('.private_segment_fixed_size','?'):>5}  lds {r.get('.group_segment_fixed_size','?')}")
