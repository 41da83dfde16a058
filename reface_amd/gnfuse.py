"""Bookkeeping that lets a GroupNorm take its statistics from the GEMMs that wrote its input (ops.fuse_groupnorm_stats).

An engine records every rf_conv_gemm launch with the [B, H, W, C] view it writes; buffers that go back to the engine's free list
are forgotten (their memory will be rewritten by something else).  ``producers(x)`` returns the recorded launches that tile x
exactly -- column slices of a concat buffer, batch halves -- or None.
"""


class ProducerTracker:
    def __init__(self):
        self.produced = []          # (ptr, pixel pitch in bytes, rows, row bytes, launch)

    def record(self, out, launch):
        if out is not None and out.dim() == 4:
            es = out.element_size()
            self.produced.append((out.data_ptr(), out.stride(2) * es, out.shape[0] * out.shape[1] * out.shape[2], out.shape[3] * es, launch))
        return launch

    def forget(self, t):
        st = t.untyped_storage()
        lo, hi = st.data_ptr(), st.data_ptr() + st.nbytes()
        self.produced = [r for r in self.produced if not (lo <= r[0] < hi)]

    def producers(self, x):
        es, pitch = x.element_size(), x.stride(2) * x.element_size()
        M, cb, base = x.shape[0] * x.shape[1] * x.shape[2], x.shape[3] * es, x.data_ptr()
        if x.stride(1) != x.shape[2] * x.stride(2) or (x.shape[0] > 1 and x.stride(0) != x.shape[1] * x.stride(1)):
            return None
        found = {}
        for ptr, p2, rows, colbytes, l in self.produced:            # later entries overwrite earlier ones
            off = ptr - base
            if p2 != pitch or off < 0:
                continue
            row0, cb0 = divmod(off, pitch)
            if row0 + rows <= M and cb0 + colbytes <= cb:
                found[(row0, cb0)] = (l, row0, rows, cb0 // es, colbytes // es)
        prods = list(found.values())
        if not prods or sum(r * c for _, _, r, _, c in prods) != M * x.shape[3]:
            return None
        return prods
