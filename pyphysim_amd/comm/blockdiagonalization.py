"""Mirror of pyphysim.comm.blockdiagonalization for the square multi-user downlink (as many transmit as
receive antennas): ``block_diagonalize``, ``calc_receive_filter`` and ``BlockDiagonalizer`` with the
reference's names, arguments and errors (comm/blockdiagonalization.py:62-118, 181-664).

The precoder's columns are singular vectors, unique up to one phase per stream; the kernels return the
representative whose largest entry per column is real and positive, the reference returns LAPACK's.  Every
quantity the reference's tests pin (block-diagonal newH, power constraints, W newH = I, singular values)
is the same; so are error statistics.  The variants with external-interference handling (WhiteningBD,
EnhancedBD) are not part of this build."""
import numpy as np

from ..engine import get_engine

__all__ = ["block_diagonalize", "calc_receive_filter", "BlockDiagonalizer"]


def block_diagonalize(mtChannel, num_users, iPu, noise_var, engine=None):
    """(newH, Ms_good) of blockdiagonalization.py:62-95."""
    return BlockDiagonalizer(num_users, iPu, noise_var, engine=engine).block_diagonalize(mtChannel)


def calc_receive_filter(newH, engine=None):
    """pinv(newH) (blockdiagonalization.py:98-117, 568-585)."""
    return BlockDiagonalizer.calc_receive_filter(newH, engine=engine)


class BlockDiagonalizer:
    """reference blockdiagonalization.py:181-664."""

    def __init__(self, num_users, iPu, noise_var, engine=None):
        self.num_users = num_users
        self.iPu = iPu
        self.noise_var = noise_var
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def _solve(self, mtChannel, waterfilling, iPu=None, noise_var=None):
        H = np.asarray(mtChannel)
        # the reference's assertion and message (blockdiagonalization.py:311-313)
        assert H.shape[0] % self.num_users == 0, ("`block_diagonalize`: Number of rows of the channel must be"
                                                  " a multiple of the number of users.")
        if H.ndim != 2 or H.shape[0] != H.shape[1]:
            raise NotImplementedError("this build block-diagonalises square channels (total transmit antennas == "
                                      "total receive antennas)")
        out = self.engine.block_diagonalize(H, self.num_users, self.iPu if iPu is None else iPu,
                                            self.noise_var if noise_var is None else noise_var, waterfilling)
        if out["skipped"][0]:
            raise np.linalg.LinAlgError("Singular matrix")
        return out

    def _calc_BD_matrix_no_power_scaling(self, mtChannel):
        """(Ms_bad, Sigma): unit-norm precoder columns and the singular values of the users' equivalent
        channels, ascending inside each user (blockdiagonalization.py:272-363)."""
        n_u = np.asarray(mtChannel).shape[1] // self.num_users
        out = self._solve(mtChannel, False, iPu=float(n_u))      # each block scaled to Frobenius norm sqrt(n_u)
        return out["Ms"][0], out["sigma"][0]

    def _perform_global_waterfilling_power_scaling(self, Ms_bad, Sigma):
        """blockdiagonalization.py:365-401."""
        P, _ = self.engine.waterfilling(np.asarray(Sigma, dtype=float) ** 2, self.num_users * self.iPu,
                                        self.noise_var)
        return np.asarray(Ms_bad) * np.sqrt(P)[None, :]

    def _perform_normalized_waterfilling_power_scaling(self, Ms_bad, Sigma):
        """blockdiagonalization.py:403-464."""
        n_u = np.asarray(Sigma).size // self.num_users
        Ms_good = self._perform_global_waterfilling_power_scaling(Ms_bad, Sigma)
        max_sqrt_P = max(np.linalg.norm(Ms_good[:, u * n_u:(u + 1) * n_u], "fro") for u in range(self.num_users))
        return Ms_good * np.sqrt(self.iPu) / max_sqrt_P

    def block_diagonalize(self, mtChannel):
        """(newH, Ms_good) with the normalised water-filling (blockdiagonalization.py:466-508)."""
        out = self._solve(mtChannel, True)
        return out["newH"][0], out["Ms"][0]

    def block_diagonalize_no_waterfilling(self, mtChannel):
        """(newH, Ms_good), every user's block at Frobenius norm sqrt(iPu) (blockdiagonalization.py:510-566)."""
        out = self._solve(mtChannel, False)
        return out["newH"][0], out["Ms"][0]

    @staticmethod
    def calc_receive_filter(newH, engine=None):
        eng = engine if engine is not None else get_engine()
        return eng.pinv(np.asarray(newH))
