"""Developer aid: resource usage and memory-op skeleton of one kernel in the gfx950 ISA listing
(`make -C topolow_amd/csrc asm` writes topolow_relax.gfx950.s).
usage: python tools/isa_report.py <mangled-name-substring> [<substring> ...] [--dump FILE]"""
import re, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
if dump: args = [a for a in args if a != dump]
s = open("/root/repo/topolow_amd/csrc/topolow_relax.gfx950.s").read()
names = [n for n in re.findall(r"^(_Z\w+):", s, re.M) if all(a in n for a in args)]
for name in names:
    i = s.index("\n" + name + ":"); j = s.index(".Lfunc_end", i)
    body = s[i:j]
    k = s.index(".amdhsa_kernel " + name); meta = s[k:s.index(".end_amdhsa_kernel", k)]
    g = lambda pat, txt: (re.search(pat, txt) or [None, "?"])[1]
    print(name[:110])
    print("  vgpr", g(r"\.amdhsa_next_free_vgpr (\d+)", meta), "sgpr", g(r"\.amdhsa_next_free_sgpr (\d+)", meta),
          "lds", g(r"\.amdhsa_group_segment_fixed_size (\d+)", meta), "scratch", g(r"; ScratchSize: (\d+)", s[j:j + 3000]),
          "occupancy", g(r"; Occupancy: (\d+)", s[j:j + 3000]), "lines", body.count("\n"))
    if dump:
        open(dump, "w").write(body)
