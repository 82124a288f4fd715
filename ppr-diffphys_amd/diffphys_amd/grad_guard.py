"""Per-parameter gradient-norm history of ``phys_model.check_grad`` (/root/reference/diffphys/dp_model.py:965-1000), batched.

The reference keeps a Python list of 0-dim GPU tensors per parameter, stacks and medians each one every iteration and
branches on a GPU scalar per parameter: ~4 launches and one host synchronisation per parameter and iteration (80 parameters).
Here the histories are the rows of ONE device tensor and an iteration is a dozen launches and a single host transfer,
with the same decisions and the same values:

  * a parameter's history grows to ``queue_length + 1`` norms, then slides;
  * once it is full, ``med`` = torch.median of all but the newest entry (the lower median of ``queue_length`` values);
    a gradient norm above ``scale_threshold * med`` is an outlier: the gradient is clipped to norm ``med``
    (``clip_grad_norm_``: scaled by ``med / (norm + 1e-6)``) and the history is left alone; otherwise the norm is pushed.

Two phases, because the reference returns BEFORE touching the histories when the global norm is over its threshold:
``plan()`` launches the device work and returns a small tensor of flags for the caller's one host transfer;
``commit()`` applies the decisions.
"""
import torch


class GradHistory:
    def __init__(self, queue_length=10, scale_threshold=5.0):
        self.L, self.scale = queue_length, scale_threshold
        self.names = None   # row order
        self.Q = None       # [P, L + 1] norms, oldest first, left-aligned
        self.fill = None    # host list: entries per row
        self._cache = {}

    # ------------------------------------------------------------------ helpers
    def _rows(self, names, device):
        if self.names is None:
            self.names = list(names)
            self.Q = torch.zeros(len(self.names), self.L + 1, device=device, dtype=torch.float32)
            self.fill = [0] * len(self.names)
        known = {n: i for i, n in enumerate(self.names)}
        new = [n for n in names if n not in known]
        if new:  # a parameter that shows up later starts an empty history, like the reference's dict
            self.names += new
            self.Q = torch.cat([self.Q, torch.zeros(len(new), self.L + 1, device=device, dtype=torch.float32)], 0)
            self.fill += [0] * len(new)
            known = {n: i for i, n in enumerate(self.names)}
        return [known[n] for n in names]

    def _const(self, key, build):
        t = self._cache.get(key)
        if t is None:
            if len(self._cache) > 64:
                self._cache.clear()
            t = self._cache[key] = build()
        return t

    # -------------------------------------------------------------------- phases
    def plan(self, names, grads):
        """names / grads of the parameters that have a gradient this iteration.  Device work only.
        Returns a dict; plan["any_outlier"] is a 0-dim float tensor (or None when no history is full yet)."""
        dev = grads[0].device
        rows = self._rows(names, dev)
        norms = torch.stack(torch._foreach_norm(grads)).float()
        all_rows = rows == list(range(len(self.names)))
        fills = [self.fill[r] for r in rows]
        has_med = [f > self.L for f in fills]
        p = {"rows": rows, "all_rows": all_rows, "norms": norms, "has_med": has_med, "grads": grads, "med": None, "outlier": None,
             "any_outlier": None}
        if any(has_med):
            Q = self.Q if all_rows else self.Q[self._const(("rows", tuple(rows)), lambda: torch.tensor(rows, device=dev))]
            med = Q[:, : self.L].median(1).values
            outlier = norms > self.scale * med
            if not all(has_med):
                outlier = outlier & self._const(("mask", tuple(has_med)), lambda: torch.tensor(has_med, device=dev))
            p["med"], p["outlier"], p["any_outlier"] = med, outlier, outlier.any().float()
        return p

    def commit(self, p, any_outlier):
        """Applies plan ``p``; ``any_outlier`` is the host value of p["any_outlier"] (False when that was None).
        Returns the list of outlier flags (host) -- all False on the common path, which needs no further transfer."""
        rows, norms, n = p["rows"], p["norms"], len(p["rows"])
        flags = [False] * n
        if any_outlier:  # read which (and by how much) in ONE transfer, clip those gradients in place with ONE multi-tensor launch
            coef = (p["med"] / (norms + 1e-6)).clamp(max=1.0)
            host = torch.stack([p["outlier"].to(coef.dtype), coef]).tolist()
            flags = [f != 0 for f in host[0]]
            hit = [i for i, f in enumerate(flags) if f]
            if hit:  # (the fp32 coefficient as a Python float is the same number: the same product as mul_ by the 0-dim tensor)
                torch._foreach_mul_([p["grads"][i] for i in hit], [host[1][i] for i in hit])
        fills = [self.fill[r] for r in rows]
        if p["all_rows"] and len(set(fills)) == 1 and (fills[0] > self.L or not any(flags)):
            # every history in the same state: one or two launches.  Outliers exist only among FULL histories (no median before), and a
            # full history stays full: the flagged rows keep theirs, the others slide -- one select over the table instead of a Python loop
            # of two launches per parameter (the reference's guard flags some parameter in most iterations of a real run: 4 022 flags in
            # run.sh's 505 iterations -- this IS the common path)
            f = fills[0]
            if f > self.L:
                slid = torch.cat([self.Q[:, 1:], norms[:, None]], 1)
                self.Q = torch.where(p["outlier"][:, None], self.Q, slid) if any(flags) else slid
            else:
                self.Q[:, f] = norms
                self.fill = [f + 1] * n
            return flags
        for i, r in enumerate(rows):  # general case (parameters come and go, or an outlier this iteration)
            if flags[i]:
                continue
            f = self.fill[r]
            if f > self.L:
                self.Q[r] = torch.cat([self.Q[r, 1:], norms[i : i + 1]])
            else:
                self.Q[r, f] = norms[i]
                self.fill[r] = f + 1
        return flags

    # ------------------------------------------------------------- compatibility
    def history(self, name):
        """the reference's ``grad_queue[name]`` as a host list (tests, debugging)"""
        r = self.names.index(name)
        return self.Q[r, : self.fill[r]].tolist()
