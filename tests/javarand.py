"""java.util.Random (the documented 48-bit LCG), so the reference's seeded tests can be re-derived."""


class JavaRandom:
    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        r = self.seed >> (48 - bits)
        if r >= 1 << (bits - 1):  # to signed int of `bits` width ... Java returns (int)(seed >>> (48 - bits))
            pass
        r &= 0xFFFFFFFF
        return r - (1 << 32) if r >= (1 << 31) else r

    def next_int(self, bound=None):
        if bound is None:
            return self.next(32)
        assert bound > 0
        r = self.next(31)
        m = bound - 1
        if (bound & m) == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):
                return r
            u = self.next(31)

    def next_long(self):
        hi = self.next(32)
        lo = self.next(32)
        v = ((hi << 32) + lo) & 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def next_float(self):
        return self.next(24) / float(1 << 24)

    def next_double(self):
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))
