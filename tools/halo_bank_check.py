"""LDS bank check of the halo-resident conv's activation fragment reads (csrc/conv_halo.hip: "Halo image"), by enumeration.

Layout: halo pixel p = hy * 18 + hx lives at byte p * 128 of the buffer; its eight 16-byte channel chunks are XOR-swizzled by (hx & 7).  A
fragment read is one ds_read_b128 per lane: lane = fq * 16 + fr reads, for tap (ky, kx), patch row py and k-half kh, pixel (py + ky, fr + kx),
logical chunk kh * 4 + fq.  ds_read_b128 is served in four NON-CONTIGUOUS 16-lane groups (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27},
{4-11, 16-19, 28-31} and the same + 32); LDS has 64 banks x 4 B = one 256-byte row, so a group is conflict-free when its 16 x 16 bytes fall on
16 different 16-byte slots of the bank row (address mod 256).  The check runs every tap, patch row, k-half and lane group, with and without
the swizzle (without it, the 128-byte pixel pitch puts lanes fr and fr + 2 on the same slot).

CPU-only; run: python tools/halo_bank_check.py"""
HP = 18


G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]


def slots(ky, kx, py, kh, group, swizzle=True):
    out = []
    for lane in GROUPS[group]:
        fr, fq = lane & 15, lane >> 4
        hy, hx = py + ky, fr + kx
        chunk = kh * 4 + fq
        if swizzle:
            chunk ^= hx & 7
        out.append(((hy * HP + hx) * 128 + chunk * 16) % 256 // 16)
    return out


def worst(swizzle):
    w = 1
    for ky in range(3):
        for kx in range(3):
            for py in range(16):
                for kh in range(2):
                    for group in range(4):
                        s = slots(ky, kx, py, kh, group, swizzle)
                        w = max(w, max(s.count(v) for v in set(s)))
    return w


if __name__ == "__main__":
    ws, wp = worst(True), worst(False)
    print(f"halo fragment reads, worst lanes per 16-byte bank slot in a 16-lane group: swizzled {ws} (1 = conflict-free), plain {wp}")
    assert ws == 1 and wp > 1
