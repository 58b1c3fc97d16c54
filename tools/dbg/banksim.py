"""Bank-conflict simulation of candidate LDS images for a [64 rows][64 bf16] tile that serves BOTH the row-read operand
(ds_read_b128: lane (r, half) reads 16 B = chunk 2c+half of row swz23(r) + 32 blk) and the transposed operand
(ds_read_b64_tr_b16: lane 4q+p of a 16-lane group reads 8 B at row r0+q, cols c0 + 16(G&1) + 4p ...)."""
import itertools
def swz23(i): return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1)
B128_GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
               [32+x for x in list(range(0,4))+list(range(12,16))+list(range(20,28))], [32+x for x in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def conflicts(addrs, nbytes, groups, nbanks=64):
    """max extra cycles: per group, per bank, number of distinct dword addresses - 1 (max over banks), summed over groups"""
    tot = 0
    for g in groups:
        bank = {}
        for l in g:
            for d in range(nbytes // 4):
                a = addrs[l] + 4 * d
                bank.setdefault((a // 4) % nbanks, set()).add(a // 4)
        tot += max(len(v) for v in bank.values()) - 1
    return tot
def off_a(row, ch):      # guide layout (a) adapted to 128-byte rows: 8-row x 32-col subtiles of 512 B, two per row group
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3))
def off_b(row, ch):      # plain 128-byte rows with an XOR on the chunk index
    return 128 * row + 16 * (ch ^ (row & 7))
def off_c(row, ch):
    return 128 * row + 16 * (ch ^ ((row >> 1) & 7))
def off_d(row, ch):      # two rows per 256 B bank row
    return 128 * row + 16 * (ch ^ (((row & 3) << 1) | ((row >> 2) & 1)))
def off_pad(row, ch):    # current row image: stride 144 B
    return 144 * row + 16 * ch
def off_pad192(row, ch):
    return 192 * row + 16 * ch
def test(name, off):
    worst_row = 0
    for blk in (0, 1):
        for c in range(4):
            addrs = {}
            for lane in range(64):
                r, half = lane & 31, lane >> 5
                addrs[lane] = off(32 * blk + swz23(r), 2 * c + half)
            worst_row = max(worst_row, conflicts(addrs, 16, B128_GROUPS))
    # also the un-swizzled row order (r instead of swz23(r))
    worst_row_plain = 0
    for blk in (0, 1):
        for c in range(4):
            addrs = {lane: off(32 * blk + (lane & 31), 2 * c + (lane >> 5)) for lane in range(64)}
            worst_row_plain = max(worst_row_plain, conflicts(addrs, 16, B128_GROUPS))
    worst_tr = 0
    for row0 in range(0, 64, 16):
        for col0 in (0, 32):
            for plus4 in (0, 4):
                addrs = {}
                for lane in range(64):
                    G, i = lane >> 4, lane & 15
                    row = row0 + 8 * (G >> 1) + (i >> 2) + plus4
                    col = col0 + 16 * (G & 1) + 4 * (i & 3)          # element column; 8 bytes = 4 elements
                    ch, within = col // 8, (col % 8) * 2
                    addrs[lane] = off(row, ch) + within
                worst_tr = max(worst_tr, conflicts(addrs, 8, [list(range(32)), list(range(32, 64))]))
    # ds_write_b128 by 256 threads: thread t writes chunk t&7 of rows t>>3 and 32 + (t>>3); groups of 8 contiguous lanes, 32 banks
    worst_w = 0
    for wave in range(4):
        addrs = {lane: off((64 * wave + lane) >> 3, lane & 7) for lane in range(64)}
        worst_w = max(worst_w, conflicts(addrs, 16, [list(range(8 * k, 8 * k + 8)) for k in range(8)], nbanks=32))
    print(f"{name:10s} row-read(swz23) extra cycles {worst_row}  row-read(plain) {worst_row_plain}  tr-read {worst_tr}  write(8-lane groups, 32 banks) {worst_w}")
for n, f in (("pad144", off_pad), ("pad192", off_pad192), ("a", off_a), ("b", off_b), ("c", off_c), ("d", off_d)):
    test(n, f)
