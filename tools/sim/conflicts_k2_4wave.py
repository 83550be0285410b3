# K2 of deform_hot.hip (4 waves, tiles of 8 x 8 x 16 output voxels, 4 voxels per lane): LDS atomic cost of the 64
# ds_add_u32 per voxel under different lane -> voxel maps and cell layouts, on the cfg2 coordinate field
# (tools/sim/field.py).  Model (profiles/r02_ubench_lds.txt): a wave's atomic goes through the LDS as 4 groups of 16
# lanes, 32 banks; a group costs the maximum over the banks of the sum, over the distinct addresses on that bank, of
# 1 (one lane), 1.5 (two lanes on the address), 0.875 k (k lanes).
import numpy as np, sys, itertools
n = 256
s = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ntile = int(sys.argv[2]) if len(sys.argv) > 2 else 200
c = np.load(f'/tmp/sim/coords_s{s}.npy').astype(np.float64)
c = np.abs(c); c = np.where(c > n - 1, 2 * (n - 1) - c, c)
st = np.floor(c).astype(np.int32) - 1
rng = np.random.default_rng(0)
tiles = [(rng.integers(0, n // 8), rng.integers(0, n // 8), rng.integers(0, n // 16)) for _ in range(ntile)]

def cost_addr(k):
    return 1.0 if k == 1 else (1.5 if k == 2 else 0.875 * k)

def group_cost(addr):
    bank = addr % 32
    best = 0.0
    for b in np.unique(bank):
        a = addr[bank == b]
        _, cnt = np.unique(a, return_counts=True)
        best = max(best, sum(cost_addr(k) for k in cnt))
    return best

# lane maps: tid (0..255) -> (z of voxel 0, y, x), voxel i at z + zstep * i
def map_shipped(t):      # x two apart inside 16 lanes, rows y and y + 2
    return t >> 7, 4 * ((t >> 6) & 1) + ((t >> 5) & 1) + 2 * ((t >> 3) & 1), 2 * (t & 7) + ((t >> 4) & 1)
def map_rows16(t):       # 16 consecutive x per 16 lanes
    return t >> 7, (t >> 4) & 7, t & 15
def map_x2_z(t):         # 8 x two apart, two z planes (z, z + 1) per 16 lanes -- voxel i at z + 2 i
    return ((t >> 3) & 1), (t >> 5) & 7, 2 * (t & 7) + ((t >> 4) & 1)
def map_x4(t):           # 4 x four apart x 4 rows (y, y+2, y+4, y+6)
    return t >> 7, 2 * ((t >> 2) & 3) + ((t >> 6) & 1), 4 * (t & 3) + ((t >> 4) & 3)
def map_x2_y4(t):        # 8 x two apart, rows y and y + 4
    return t >> 7, ((t >> 5) & 3) + 4 * ((t >> 3) & 1), 2 * (t & 7) + ((t >> 4) & 1)
def map_x2_y1(t):        # 8 x two apart, rows y and y + 1
    return t >> 7, 2 * ((t >> 5) & 3) + ((t >> 3) & 1), 2 * (t & 7) + ((t >> 4) & 1)
def map_checker(t):      # 8 even x of row y + 8 odd x of row y + 1 per 16 lanes
    j = t & 15; row = (t >> 4) & 7; yy = (row & ~1) + (j >> 3); xx = 2 * (j & 7) + ((row ^ (j >> 3)) & 1)
    return t >> 7, yy, xx
def map_rows16_zalt(t):  # 16 consecutive x per 16 lanes; the wave's 4 groups are 4 different z (not 4 y)
    return (t >> 4) & 1, ((t >> 5) & 1) + 2 * (t >> 6), t & 15
MAPS = dict(checker=map_checker, shipped=map_shipped, rows16=map_rows16, x2_z=map_x2_z, x4=map_x4, x2_y4=map_x2_y4, x2_y1=map_x2_y1)
ZSTEP = dict(x2_z=2)

def pitch_shipped(e):
    return 8 if e <= 8 else 24 if e <= 24 else 40 if e <= 40 else 56
LAYOUTS = {
    'P=32 PS32': (lambda e: 32 if e <= 32 else 64, lambda by, P: by * P),
    'P=24/40 PS=0m32': (lambda e: 24 if e <= 24 else 40 if e <= 40 else 56, lambda by, P: by * P + ((0 - by * P) % 32)),
    'P=8odd (shipped)': (pitch_shipped, lambda by, P: by * P),
    'P=8odd PS+8m32': (pitch_shipped, lambda by, P: by * P + ((8 - by * P) % 32)),
    'P=8odd PS+16m32': (pitch_shipped, lambda by, P: by * P + ((16 - by * P) % 32)),
    'P=ext|1': (lambda e: e | 1, lambda by, P: by * P),
    'P=32': (lambda e: 32 if e <= 32 else 64, lambda by, P: by * P),
    'P=32+1': (lambda e: 33 if e <= 33 else 65, lambda by, P: by * P),
    'P=16odd': (lambda e: 16 if e <= 16 else 48, lambda by, P: by * P),
    'P=ext+pad to 4m8': (lambda e: e + ((4 - e) % 8), lambda by, P: by * P),
}
for mname, mp in MAPS.items():
    zs = ZSTEP.get(mname, 2)
    lanes = np.array([mp(t) for t in range(256)])
    # check the map covers the tile
    cover = set()
    for t in range(256):
        for i in range(4):
            cover.add((lanes[t, 0] + zs * i if mname != 'x2_z' else lanes[t, 0] + 2 * i, lanes[t, 1], lanes[t, 2]))
    assert len(cover) == 1024, (mname, len(cover))
    for lname, (Pf, PSf) in LAYOUTS.items():
        tot = 0.0; cnt = 0; cells = 0
        for (tz, ty, tx) in tiles:
            s3 = st[:, tz * 8:tz * 8 + 8, ty * 8:ty * 8 + 8, tx * 16:tx * 16 + 16]
            lo = s3.reshape(3, -1).min(1); hi = s3.reshape(3, -1).max(1) + 4
            ext = hi - lo; r = s3 - lo[:, None, None, None]
            P = Pf(ext[2]); PS = PSf(ext[1], P)
            cells += ext[0] * PS
            base = r[0] * PS + r[1] * P + r[2]
            for i in range(4):
                a = base[lanes[:, 0] + 2 * i, lanes[:, 1], lanes[:, 2]]
                for g in range(16):
                    tot += group_cost(a[g * 16:(g + 1) * 16]); cnt += 1

        print(f'sigma {s} map {mname:8s} layout {lname:18s}: cycles per 16-lane group {tot / cnt:.2f}  (x 256 groups per voxel-wave = {tot / cnt * 256:.0f}; ideal 256)  mean cells {cells / len(tiles):.0f}')
