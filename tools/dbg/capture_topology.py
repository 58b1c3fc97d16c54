"""Which cross-stream wait patterns survive hipGraph capture on this ROCm?  Each variant runs in its own process."""
import subprocess, sys
VARIANTS = {
    "fork_join": "B<A C<A A<B A<C",
    "side_waits_side": "B<A C<A kB C<B kC A<B A<C",
    "both_directions": "B<A C<A kB C<B kC kB B<C kB A<B A<C",
    "both_plus_back": "B<A C<A kB C<B kC kB B<C kB C<B kC A<B A<C",
    "via_origin": "B<A C<A kB A<B C<A kC kB B<C kB C<B kC A<B A<C",
    "join_back_to_forker": "B<A kB C<B kC B<C kB A<B",
    "sibling_join": "B<A C<A kB kC B<C kB A<B",
    "third_stream_feeds_siblings": "B<A C<A D<A kD B<D C<D kB kC B<C kB A<B A<C A<D",
    "third_stream_unjoined": "B<A C<A D<A kD B<D C<D kB kC B<C kB A<B A<C",
}
CHILD = r'''
import sys, torch
ops = sys.argv[1].split()
dev = torch.device("cuda", 0)
S = {k: torch.cuda.Stream(dev) for k in "ABCD"}
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=S["A"]):
    x.add_(1)
    for op in ops:
        if op[0] == "k":
            with torch.cuda.stream(S[op[1]]):
                x.add_(1)
        else:
            S[op[0]].wait_stream(S[op[2]])
    x.add_(1)
g.replay(); torch.cuda.synchronize()
print("ok", float(x[0]))
'''
for name, ops in VARIANTS.items():
    r = subprocess.run([sys.executable, "-c", CHILD, ops], capture_output=True, text=True, timeout=120)
    print(name, "->", r.returncode, (r.stdout.strip() or r.stderr.strip()[-200:]), flush=True)
