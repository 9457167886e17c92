"""Noise models of the MC path - API mirror of noise_model.py, evaluated on the GPU.

    noise_function             stateful RNG wrapper: a call merges its kwargs into the stored ones and then
                               draws (noise_model.py:21-46) - so ``rng(scale=s)`` sets sigma AND burns a draw.
    noise_model_base           chain/ring XX Hamiltonian `HH`, one-hot controls `CC`, default Gaussian rng
                               (noise_model.py:71-95, :114-115) and `evaluate_noisy_fidelity` (:98-109).
    structured_perturbation    3N draws per sample in the order (g0_i, g1_i, g2_i) (noise_model.py:122-147).

What differs from the reference is only WHERE the arithmetic runs: a sample is described by its 3N draws
(never by a dense N x N perturbation matrix) and fidelities come from `rc_mc_fidelity_f64`.  The batched
entry `fidelity_batch` draws a whole (C, K, N, 3) tensor with ONE generator call, which consumes numpy's
legacy stream exactly like the reference's 3 N C K scalar calls (same values, same order).
"""
from __future__ import annotations

import numpy as np

from . import backend


class noise_function:
    def __init__(self, generator, **args):
        self.generator = generator
        self.args = args

    def __call__(self, **extraargs):
        self.args.update(extraargs)          # sticky, as in the reference (noise_model.py:42-44)
        return self.generator(**self.args)

    def draw_block(self, shape):
        """`shape` draws with the current arguments WITHOUT making `size` sticky.  Generators that take
        ``size=`` (numpy's do) are called once; anything else is called element by element in C order."""
        try:
            block = self.generator(**{**self.args, "size": tuple(shape)})
            block = np.asarray(block, dtype=np.float64)
            if block.shape == tuple(shape):
                return block
        except TypeError:
            pass
        flat = np.empty(int(np.prod(shape)), dtype=np.float64)
        for i in range(flat.size):
            flat[i] = self.generator(**self.args)
        return flat.reshape(shape)


class noise_model_base:
    """Same constructor, attributes and methods as the reference class (noise_model.py:50-115)."""

    def __init__(self, Nspin: int = 5, inspin: int = 0, outspin: int = 2, noise: float = 0.02,
                 topo: str = "chain", rng: noise_function = None, device=None):
        self.Nspin = Nspin
        self.inspin = inspin
        self.outspin = outspin
        self.noise = noise
        self.rng = self.default_gaussian_noise_generator(scale=self.noise) if rng is None else rng
        self.HH = np.zeros((Nspin, Nspin), dtype=np.complex128)
        hop = np.arange(1, Nspin)
        self.HH[hop - 1, hop] = 1
        self.HH[hop, hop - 1] = 1
        if topo == "ring":
            self.HH[Nspin - 1, 0] = 1
            self.HH[0, Nspin - 1] = 1
        self.CC = self.controls()
        self.device = device            # None = torch's current device at call time (one process per GPU)

    def controls(self):
        return [np.diag((np.arange(self.Nspin) == k).astype(np.float64)) for k in range(self.Nspin)]

    def default_gaussian_noise_generator(self, **genargs):
        return noise_function(np.random.normal, **genargs)

    # -- static part of the Hamiltonian as the kernel wants it ------------------------------------
    def _static_terms(self):
        """(h0_diag, h0_offdiag, ring, imag_offdiag) read back from the public `HH` attribute, so that a
        caller who edits `HH` (e.g. adds the XXZ diagonal of qnewton.py:148-150) is honoured."""
        n = self.Nspin
        H = np.asarray(self.HH)
        hop = np.arange(1, n)
        allowed = np.zeros((n, n), dtype=bool)
        allowed[np.arange(n), np.arange(n)] = True
        allowed[hop, hop - 1] = allowed[hop - 1, hop] = True
        ring = bool(n > 2 and (H[n - 1, 0] != 0 or H[0, n - 1] != 0))
        if ring:
            allowed[n - 1, 0] = allowed[0, n - 1] = True
            if not (H[n - 1, 0] == 1 and H[0, n - 1] == 1):
                raise NotImplementedError("ring closure other than J = 1 is not supported by the kernel")
        if (H[~allowed] != 0).any() or not np.allclose(H, H.conj().T, atol=0, rtol=0):
            raise NotImplementedError("HH must be Hermitian nearest-neighbour (chain or ring)")
        diag = H.diagonal().real.copy()
        if (H.diagonal().imag != 0).any():
            raise NotImplementedError("HH must have a real diagonal")
        lower = H[hop, hop - 1]
        return diag, lower.real.copy(), ring, lower.imag.copy()

    # -- sampling ---------------------------------------------------------------------------------
    def draw_samples(self, n_controllers: int, n_draws: int) -> np.ndarray:
        """(C, K, N, 3) draws in the reference's consumption order (controller, draw, site, slot)."""
        raise NotImplementedError

    def perturbation(self) -> np.ndarray:
        raise NotImplementedError

    # -- evaluation -------------------------------------------------------------------------------
    def fidelity_from_draws(self, controllers, draws, kernel: str = "auto", out=None):
        """(C, N+1) controllers x (C, K, N, 3) draws -> (C, K) fidelities on the GPU (`out`: preallocated result)."""
        diag, off, ring, imag = self._static_terms()
        if imag.any():
            if backend._is_torch(draws):
                draws = draws.clone()
                import torch
                draws[..., 1:, 2] += torch.as_tensor(imag, device=draws.device)
            else:
                draws = np.array(draws, dtype=np.float64)
                draws[..., 1:, 2] += imag
        return backend.mc_fidelity(controllers, draws, self.Nspin, self.inspin, self.outspin, h0_diag=diag,
                                   h0_offdiag=off, ring=ring, device=self.device, kernel=kernel, out=out)

    def fidelity_batch(self, controllers, n_draws: int, ham_noisy: bool = True):
        """K noisy evaluations of every controller row; consumes the rng like C*K reference calls."""
        ctrl = np.asarray(controllers, dtype=np.float64).reshape(-1, self.Nspin + 1)
        if ham_noisy:
            draws = self.draw_samples(ctrl.shape[0], n_draws)
        else:
            draws = np.zeros((ctrl.shape[0], n_draws, self.Nspin, 3))
        return self.fidelity_from_draws(ctrl, draws)

    # -- optimiser-side objective over a FIXED set of perturbed Hamiltonians ---------------------------------
    def fixed_perturbation_set(self, size: int, real_only: bool = True) -> np.ndarray:
        """(size, N, 3) draws of a fixed Hamiltonian set, consuming numpy's global stream like the reference's
        `randHset_constructor` (qnewton.py:122-137, after its `np.random.seed(4)`): with `real_only` the
        perturbation is the optimiser's real one (qnewton.py:366-379: per site a diagonal and a coupling draw, no
        imaginary part), otherwise the MC path's three draws per site."""
        sigma = float(self.rng.args.get("scale", self.noise))
        if real_only:
            two = np.random.normal(scale=sigma, size=(size, self.Nspin, 2))
            return np.concatenate([two, np.zeros((size, self.Nspin, 1))], axis=2)
        return np.random.normal(scale=sigma, size=(size, self.Nspin, 3))

    def randHset_constructor(self, train_size: int = 1000, test_size: int = 10000):
        """The optimiser's fixed training / test Hamiltonian sets (qnewton.py:122-137): re-seeds numpy's global stream
        with 4 - as the reference does - and draws `train_size` then `test_size` real perturbations.  Returned as draw
        sets (size, N, 3) in the kernel layout instead of dense matrices."""
        np.random.seed(4)
        return self.fixed_perturbation_set(train_size), self.fixed_perturbation_set(test_size)

    def fidelity_fixed_set(self, controllers, draw_set):
        """(C, R) fidelities of C controllers on ONE set of R perturbations (no replication of the set)."""
        ctrl = np.asarray(controllers, dtype=np.float64).reshape(-1, self.Nspin + 1)
        draw_set = np.ascontiguousarray(draw_set, dtype=np.float64).reshape(1, -1, self.Nspin, 3)
        return self.fidelity_from_draws(ctrl, draw_set)

    def fidelity_ss_av(self, controllers, draw_set, reps=None):
        """Mean fidelity over a fixed set for every controller - the reference's `fidelity_ss_av` (qnewton.py:426-444)
        batched over controllers: `reps` = first Hamiltonians of the set to use (the reference's `test=False` branch
        takes reps = 10 of the training set), None = the whole set (its `test=True` branch)."""
        from . import backend as _be
        draw_set = np.asarray(draw_set, dtype=np.float64).reshape(-1, self.Nspin, 3)
        if reps is not None:
            draw_set = draw_set[: int(reps)]
        fid = self.fidelity_fixed_set(controllers, draw_set)
        return 1.0 - _be.reduce_metrics(fid, q_thresholds=(), overlapped=False)["rim1"][0]

    # -- the scalar API -----------------------------------------------------------------------------------------------
    # Reference-style callers loop `for b in range(K): f += nm.evaluate_noisy_fidelity(cont, ham_noisy=True)`
    # (gen_fig_8_arim_fcall_scaling.py:121-132).  One GPU launch + sync per sample costs what the reference's CPU
    # evaluation costs, and there is no CPU path to fall back to.  So a call LOOKS AHEAD: after drawing its own 3N
    # perturbations from numpy's stream - exactly what the reference does at this point - it also draws those of the
    # next B - 1 samples, evaluates all B in one launch, and puts the generator back to where it was after ITS sample.
    # A following call again draws its 3N values from the live stream and compares them with the ones the block was
    # computed from: equal draws (and same controller, sigma, Hamiltonian) => the stored fidelity IS this sample's
    # fidelity; anything else (somebody drew from numpy in between, `rng(scale=...)` burned a draw, another controller)
    # => the values just drawn are still this sample's draws, a new block starts from them.  `np.random`'s state is the
    # reference's at every call boundary, bit for bit, without ever being inspected.  B grows 8 -> 1024 on hits.
    _LOOKAHEAD_MAX = 1024

    def _lookahead_usable(self) -> bool:
        return type(self).draw_samples is structured_perturbation.draw_samples and backend.legacy_stream_usable(self.rng)

    def _lookahead_eval(self, x: np.ndarray) -> float:
        sigma = float(self.rng.args.get("scale", self.noise))
        shape = (self.Nspin, 3)
        mine = np.random.normal(scale=sigma, size=shape)           # this sample's draws: consumed like the reference does
        la = self.__dict__.get("_la")
        sig = (x.tobytes(), sigma, np.asarray(self.HH).tobytes(), self.inspin, self.outspin)   # HH is public and mutable
        if la is not None and la["sig"] == sig and la["i"] < len(la["fid"]) and np.array_equal(mine, la["draws"][la["i"]]):
            la["i"] += 1
            return float(la["fid"][la["i"] - 1])
        block = 8 if la is None or la["sig"] != sig else min(self._LOOKAHEAD_MAX, 2 * len(la["fid"]))
        here = np.random.get_state()
        draws = np.empty((block,) + shape)
        draws[0] = mine
        draws[1:] = np.random.normal(scale=sigma, size=(block - 1,) + shape)       # the following samples' draws
        np.random.set_state(here)                                  # ... which the reference has not consumed yet
        fid = np.asarray(self.fidelity_from_draws(x, draws[None]))[0]
        self._la = {"sig": sig, "fid": fid, "draws": draws, "i": 1}
        return float(fid[0])

    def evaluate_noisy_fidelity(self, x, ham_noisy: bool = False):
        """One sample, reference signature (noise_model.py:98-109)."""
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(1, -1)[:, : self.Nspin + 1])
        if ham_noisy and self._lookahead_usable():
            return self._lookahead_eval(x)
        return float(self.fidelity_batch(x, 1, ham_noisy)[0, 0])


class structured_perturbation(noise_model_base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def draw_samples(self, n_controllers: int, n_draws: int) -> np.ndarray:
        return self.rng.draw_block((n_controllers, n_draws, self.Nspin, 3))

    def perturbation(self) -> np.ndarray:
        """Dense matrix form of ONE sample (API parity with noise_model.py:122-147; the hot path never
        materialises it).  Consumes 3N draws."""
        g = self.draw_samples(1, 1)[0, 0]
        n = self.Nspin
        z = np.zeros((n, n), dtype=np.complex128)
        z[np.arange(n), np.arange(n)] = g[:, 0]
        lo = np.arange(1, n)
        z[lo, lo - 1] = g[1:, 1] + 1j * g[1:, 2]
        z[lo - 1, lo] = g[1:, 1] - 1j * g[1:, 2]
        return z


class directional_perturbation(noise_model_base):
    """One random element pair of the Hamiltonian perturbed per sample (noise_model.py:150-201).

    Per sample the reference consumes ``np.random.randint(0, len(directions))`` and then ``rng(size=2)`` (which
    also makes ``size=2`` sticky on the generator, as there).  For a bond direction the perturbation is Hermitian;
    for a diagonal direction (i, i) the second assignment ``z[i,i] = a - ib`` overwrites the first, so the
    Hamiltonian gets a complex diagonal entry and is NOT Hermitian - evaluated by the dense Pade-expm kernel
    (`backend.mc_fidelity_nonhermitian`), like every sample of this model.
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        n = self.Nspin
        self.directions = [(0, 0), (n - 1, n - 1)]
        for d in range(1, n - 1):
            for o in (-1, 0, 1):
                self.directions.append((d, d + o))
        self.directions += [(0, 1), (1, 0), (n - 2, n - 1), (n - 1, n - 2)]

    def _draw_one(self):
        idx = np.random.randint(low=0, high=len(self.directions))
        nval = self.rng(size=2)
        return self.directions[idx], float(nval[0]), float(nval[1])

    def perturbation(self) -> np.ndarray:
        (p, q), a, b = self._draw_one()
        z = np.zeros((self.Nspin, self.Nspin), dtype=np.complex128)
        z[p, q] = a + 1j * b
        z[q, p] = a - 1j * b
        return z

    def _draw_indices(self, n: int):
        """(direction index [n], (a, b) [n, 2]) of n samples, consuming numpy's global stream sample by sample exactly
        like n calls of the reference's `perturbation()`: `np.random.randint(0, len(directions))`, then `rng(size=2)`.
        With the default generator the whole sequence is produced by the library's host-side emulation of the legacy
        stream (`rc_directional_draws_legacy`: bit-identical indices, normals and generator state, ~100x the Python
        loop); any other generator takes the sample-by-sample loop."""
        import ctypes
        from . import _lib
        rng = self.rng
        if not self._plain_legacy() or n == 0:
            idx = np.empty(n, dtype=np.int32)
            ab = np.empty((n, 2))
            for i in range(n):
                idx[i] = np.random.randint(low=0, high=len(self.directions))
                ab[i] = self.rng(size=2)
            return idx, ab
        lib = _lib.load()
        name, key, pos, has_gauss, cached = np.random.get_state()
        st = _lib.Mt19937State()
        ctypes.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
        st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(cached)
        idx = np.empty(n, dtype=np.int32)
        ab = np.empty((n, 2))
        sigma = float(rng.args.get("scale", self.noise))
        rc = lib.rc_directional_draws_legacy(ctypes.byref(st), n, len(self.directions), sigma,
                                             ctypes.c_void_p(idx.ctypes.data), ctypes.c_void_p(ab.ctypes.data))
        if rc != 0:
            raise ValueError("rc_directional_draws_legacy failed")
        np.random.set_state(("MT19937", np.frombuffer(st.key, dtype=np.uint32).copy(), int(st.pos), int(st.has_gauss),
                             float(st.gauss)))
        rng.args["size"] = 2                          # `rng(size=2)` is sticky on the generator, as in the reference
        return idx, ab

    def draw_samples(self, n_controllers: int, n_draws: int):
        """(draws (C, K, N, 3), diag_imag (C, K, N)) in the kernel layout, consuming the RNG sample by sample in
        (controller, draw) order exactly like C*K calls of the reference's `perturbation()`."""
        idx, ab = self._draw_indices(n_controllers * n_draws)
        return self._layout(idx, ab, n_controllers, n_draws)

    def _layout(self, idx, ab, n_controllers: int, n_draws: int):
        """The dense kernel layout (draws (C, K, N, 3), diag_imag (C, K, N)) of samples given as (direction index, a, b)."""
        n = self.Nspin
        total = n_controllers * n_draws
        dirs = np.asarray(self.directions, dtype=np.int64)            # (3N - 2, 2)
        p, q = dirs[idx, 0], dirs[idx, 1]
        a, b = ab[:, 0], ab[:, 1]
        draws = np.zeros((total, n, 3))
        imag = np.zeros((total, n))
        s = np.arange(total)
        diag = p == q
        draws[s[diag], p[diag], 0] = a[diag]          # z[p,p] = a + ib, then overwritten by a - ib (noise_model.py:196-199)
        imag[s[diag], p[diag]] = -b[diag]
        low = p == q + 1                              # z[p][p-1] = a + ib: the lower element of bond p
        draws[s[low], p[low], 1], draws[s[low], p[low], 2] = a[low], b[low]
        up = p == q - 1                               # z[p][p+1] = a + ib -> lower element z[q][p] = a - ib
        draws[s[up], q[up], 1], draws[s[up], q[up], 2] = a[up], -b[up]
        return draws.reshape(n_controllers, n_draws, n, 3), imag.reshape(n_controllers, n_draws, n)

    # The scalar API of this model looks ahead like the structured model's (noise_model_base._lookahead_eval): a call draws its
    # own sample from numpy's stream exactly as the reference does (randint, then rng(size=2)), plus - through the bit-identical
    # host emulation of that consumption - the next B - 1 samples, evaluates all B at once and puts the generator back; the
    # following calls compare what they draw with what the block was computed from.  (Round 3: one launch + sync per sample.)
    def _lookahead_usable(self) -> bool:
        return self._plain_legacy()

    def _lookahead_eval(self, x: np.ndarray) -> float:
        sigma = float(self.rng.args.get("scale", self.noise))
        idx0 = int(np.random.randint(low=0, high=len(self.directions)))      # this sample, consumed like the reference does
        ab0 = np.asarray(self.rng(size=2), dtype=np.float64)
        la = self.__dict__.get("_la")
        sig = (x.tobytes(), sigma, np.asarray(self.HH).tobytes(), self.inspin, self.outspin)
        if la is not None and la["sig"] == sig and la["i"] < len(la["fid"]) and idx0 == la["idx"][la["i"]] \
                and np.array_equal(ab0, la["ab"][la["i"]]):
            la["i"] += 1
            return float(la["fid"][la["i"] - 1])
        block = 8 if la is None or la["sig"] != sig else min(self._LOOKAHEAD_MAX, 2 * len(la["fid"]))
        here = np.random.get_state()
        idx_n, ab_n = self._draw_indices(block - 1)                # the following samples' draws ...
        np.random.set_state(here)                                  # ... which the reference has not consumed yet
        idx = np.concatenate([[idx0], idx_n]).astype(np.int32)
        ab = np.concatenate([ab0[None, :], ab_n], axis=0)
        draws_t, imag = self._layout(idx, ab, 1, block)
        fid = np.asarray(self._fidelity_from_layout(x, draws_t, imag))[0]
        self._la = {"sig": sig, "fid": fid, "idx": idx, "ab": ab, "i": 1}
        return float(fid[0])

    def _plain_legacy(self) -> bool:
        rng = self.rng
        return (getattr(rng, "generator", None) is np.random.normal and set(rng.args) <= {"scale", "loc", "size"}
                and float(rng.args.get("loc", 0.0)) == 0.0 and rng.args.get("size", 2) == 2)

    def fidelity_batch(self, controllers, n_draws: int, ham_noisy: bool = True, draws: str = "auto"):
        """(C, K) fidelities.  A bond direction is a Hermitian sample of the ordinary draw layout: ALL samples first go
        through the fast chain / ring kernels (imaginary diagonal ignored); the diagonal directions - a complex diagonal
        entry, non-Hermitian - are then recomputed by the dense Pade-expm kernel on a compacted list and put in place.

        draws="device" (what "auto" picks for the default generator on a GPU): the RNG consumption itself runs on the
        GPU (`backend.directional_draws_device`), 20 bytes per sample are all that exists of a sample before the layout
        is expanded ON the device, and only the (C, K) result crosses PCIe.  draws="host": the bit-identical host
        emulation / the sample-by-sample loop for custom generators, layout built with NumPy (round 2)."""
        ctrl = np.asarray(controllers, dtype=np.float64).reshape(-1, self.Nspin + 1)
        if draws not in ("auto", "device", "host"):
            raise ValueError("draws must be 'auto', 'device' or 'host'")
        total = ctrl.shape[0] * n_draws
        # "auto": the device pipeline pays a few launches and two synchronisations - worth it from a few thousand samples
        # on; the scalar API (one sample per call) keeps the host emulation
        # ("auto" also asks that the device-continued normals ARE NumPy's on this host - backend.legacy_device_exact)
        if ham_noisy and self._plain_legacy() and (draws == "device" and total > 0 or
                                                   draws == "auto" and total >= 2048 and backend.legacy_device_exact()):
            return self._fidelity_batch_device(ctrl, n_draws)
        if draws == "device" and ham_noisy:
            raise ValueError("draws='device' needs the default generator (np.random.normal)")
        if ham_noisy:
            draws_t, imag = self.draw_samples(ctrl.shape[0], n_draws)
        else:
            draws_t, imag = np.zeros((ctrl.shape[0], n_draws, self.Nspin, 3)), None
        return self._fidelity_from_layout(ctrl, draws_t, imag)

    def _fidelity_from_layout(self, ctrl, draws_t, imag):
        """(C, K) fidelities of samples in the dense layout: everything through the fast Hermitian kernels, the samples with
        an imaginary diagonal entry (diagonal directions) recomputed by the non-Hermitian entry and put in place."""
        diag, off, ring, imag_off = self._static_terms()
        if imag_off.any():
            draws_t[..., 1:, 2] += imag_off
        fid = np.asarray(backend.mc_fidelity(ctrl, draws_t, self.Nspin, self.inspin, self.outspin, h0_diag=diag, h0_offdiag=off,
                                             ring=ring, device=self.device))
        if imag is not None:
            cs, ks = np.nonzero(imag.any(axis=2))                  # the non-Hermitian samples
            if cs.size:
                sub = backend.mc_fidelity_nonhermitian(ctrl[cs], draws_t[cs, ks][:, None], imag[cs, ks][:, None], self.Nspin,
                                                       self.inspin, self.outspin, h0_diag=diag, h0_offdiag=off, ring=ring,
                                                       device=self.device)
                fid[cs, ks] = np.asarray(sub)[:, 0]
        return fid

    def _fidelity_batch_device(self, ctrl: np.ndarray, n_draws: int) -> np.ndarray:
        """The device-resident pipeline of `fidelity_batch`: (index, a, b) per sample from the GPU-side continuation of
        numpy's stream, the (C, K, N, 3) layout scattered from them on the device, the fast kernels on everything, the
        Pade-expm kernel on the compacted diagonal-direction samples."""
        import torch
        n, C, K = self.Nspin, ctrl.shape[0], n_draws
        total = C * K
        sigma = float(self.rng.args.get("scale", self.noise))
        idx, ab = backend.directional_draws_device(total, len(self.directions), sigma, device=self.device)
        self.rng.args["size"] = 2                     # `rng(size=2)` is sticky on the generator, as in the reference
        dev = idx.device
        diag, off, ring, imag_off = self._static_terms()
        if not ring and n <= 12 and not imag_off.any():
            # round 4: ONE library call from the 20 bytes per sample to the fidelities - class partition, real tridiagonal
            # routes for the bond directions, complex symmetric QL for the diagonal ones (k_directional.inc.h); nothing of the
            # (C, K, N, 3) layout is ever built
            fid = backend.mc_fidelity_directional(torch.as_tensor(ctrl, device=dev), idx, ab, n, self.inspin, self.outspin, K,
                                                  h0_diag=diag, h0_offdiag=off)
            return self._to_host(fid)
        # rings, N > 12, complex static couplings: the dense layout, scattered on the device
        dirs = torch.as_tensor(np.asarray(self.directions, dtype=np.int64), device=dev)       # (3N - 2, 2)
        pq = dirs[idx.long()]
        p, q = pq[:, 0], pq[:, 1]
        a, b = ab[:, 0], ab[:, 1]
        draws = torch.zeros((total, n, 3), dtype=torch.float64, device=dev)
        s = torch.arange(total, device=dev)
        is_diag = p == q
        sd = s[is_diag]
        draws[sd, p[is_diag], 0] = a[is_diag]         # z[p,p] = a + ib, then overwritten by a - ib (noise_model.py:196-199)
        low = p == q + 1                              # z[p][p-1] = a + ib: the lower element of bond p
        draws[s[low], p[low], 1] = a[low]
        draws[s[low], p[low], 2] = b[low]
        up = p == q - 1                               # z[p][p+1] = a + ib -> lower element z[q][p] = a - ib
        draws[s[up], q[up], 1] = a[up]
        draws[s[up], q[up], 2] = -b[up]
        if imag_off.any():
            draws[:, 1:, 2] += torch.as_tensor(imag_off, device=dev)
        ctrl_t = torch.as_tensor(ctrl, device=dev)
        fid = backend.mc_fidelity(ctrl_t, draws.view(C, K, n, 3), self.Nspin, self.inspin, self.outspin, h0_diag=diag,
                                  h0_offdiag=off, ring=ring)
        if sd.numel():
            imag = torch.zeros((sd.numel(), 1, n), dtype=torch.float64, device=dev)
            imag[torch.arange(sd.numel(), device=dev), 0, p[is_diag]] = -b[is_diag]
            sub = backend.mc_fidelity_nonhermitian(ctrl_t[sd // K], draws[sd][:, None], imag, self.Nspin, self.inspin,
                                                   self.outspin, h0_diag=diag, h0_offdiag=off, ring=ring)
            fid.view(-1)[sd] = sub[:, 0]
        return self._to_host(fid)

    def _to_host(self, fid) -> np.ndarray:
        """(C, K) device tensor -> NumPy through a pinned staging buffer kept on the instance (pageable D2H of 8 MB runs at
        a third of the pinned rate)."""
        import torch
        # torch's caching HOST allocator: the block comes from its pool after the first call, belongs to the returned array
        # (numpy keeps the tensor alive) and goes back to the pool with it - no staging copy (8 MB: 0.35 ms of memcpy)
        host = torch.empty(fid.shape, dtype=torch.float64, pin_memory=True)
        host.copy_(fid, non_blocking=True)
        torch.cuda.current_stream(fid.device).synchronize()
        return host.numpy()
