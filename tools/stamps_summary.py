"""Keep the last stamp report per (kernel, shape, wave kind) of a conv_probe run with BC_HALO_STAMPS / BC_WREG_STAMPS (stdin or file)."""
import sys
last = {}
for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if "stamps]" in l:
        last[l.split("|")[0]] = l.strip()
    elif l.startswith("B"):
        print(l.strip()[:300])
for k in sorted(last):
    print(last[k][:330])
