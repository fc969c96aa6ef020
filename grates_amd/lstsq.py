"""
Block-banded normal equations on the device: the reference's vector-autoregressive constraint models, sparse
`BlockMatrix` and `NormalEquations` ("Kalman smoother", grates/lstsq.py:12-1149) with the same class and method
names, argument meaning and error behaviour.  The host keeps the reference's loops over non-zero blocks; every block
lives in HBM as its own fp64 tensor and every block operation is a libshg call:

    block products                  -> shg_gemm  (fp64 MFMA, transposes / alpha / beta, accumulates in place)
    scipy.linalg.cholesky           -> shg_potrf (blocked right-looking upper Cholesky)
    solve_triangular / inv of a
    diagonal factor block           -> shg_trtri once per diagonal block (cached), then shg_gemm
    block scaling / accumulation    -> shg_axpby

Vectors may be passed as NumPy arrays (results come back as NumPy arrays, like the reference) or as device tensors
(results stay on the device).  There is no CPU fallback.

Not built (outside the smoother path): UnscentedTransformSymmetric, teigh, trsvd, robust_least_squares,
AutoregressiveModel.from_transformed_coefficients (pseudo-inverse).
"""

import numpy as np

from . import engine


def _is_tensor(x):
    return engine._torch().is_tensor(x)


def _dev(x):
    """fp64 device tensor (copy) of an ndarray / tensor"""
    torch = engine.require_gpu()
    if torch.is_tensor(x):
        return x.to(device=engine.device(), dtype=torch.float64).clone(memory_format=torch.contiguous_format)   # dense row-major, whatever the strides of x
    return engine.to_device(np.asarray(x, dtype=np.float64))


def _zeros(shape):
    torch = engine.require_gpu()
    return torch.zeros(tuple(int(s) for s in shape), dtype=torch.float64, device=engine.device())


def _dot(a, b):
    """sum of the element-wise product of two [n, k] device tensors (trace of a^T b on the MFMA GEMM)"""
    return float(np.trace(engine.to_host(engine.gemm(a, b, transa=True))))


def _like_input(result, template):
    return result if _is_tensor(template) else engine.to_host(result)


class AutoregressiveModel:
    """VAR(p) model x_t = sum_k B_k x_(t-k) + w_t given by its coefficient matrices B_1 .. B_p and the covariance matrix of
    the white noise w (grates/lstsq.py:12-247)."""

    def __init__(self, coefficients, covariance_matrix):
        # an ndarray of stacked matrices is split along its first axis (upstream behaviour); lists / tuples are kept as given
        self.__coefficients = tuple(coefficients) if isinstance(coefficients, np.ndarray) else coefficients
        self.__covariance_matrix = covariance_matrix
        self.__normal_equation = None           # BlockMatrix of the pseudo-observation normals, built on first use

    @property
    def dimension(self):
        return self.__covariance_matrix.shape[0]

    @property
    def order(self):
        return len(self.__coefficients)

    @property
    def white_noise_covariance(self):
        return self.__covariance_matrix

    @property
    def coefficients(self):
        return self.__coefficients

    def order_one_representation(self):
        """grates/lstsq.py:81-99"""
        if self.order == 1:
            return self
        B = np.eye(self.dimension * self.order)
        for k in range(self.order):
            B[0:self.dimension, k * self.dimension:(k + 1) * self.dimension] = np.asarray(self.__coefficients[k]).copy()
        Q = np.zeros(B.shape)
        Q[0:self.dimension, 0:self.dimension] = np.asarray(self.__covariance_matrix).copy()
        return AutoregressiveModel(B, Q)

    @staticmethod
    def from_covariance_function(covariance_function):
        """Yule-Walker equations solved with the block Cholesky factorisation on the device (grates/lstsq.py:127-167)."""
        lags = tuple(covariance_function)                      # Sigma_0 .. Sigma_p (an ndarray is split along its first axis)
        p = len(lags) - 1
        if p == 0:
            return AutoregressiveModel((), lags[0])
        d = lags[0].shape[0]
        bounds = list(range(0, (p + 1) * d, d))               # p blocks of size d
        # block Toeplitz system of the Yule-Walker equations, upper blocks only: T[r, c] = Sigma_(c - r)^T, rhs rows = Sigma_(r + 1)
        coefficient_matrix = BlockMatrix(bounds, bounds)
        for r in range(p):
            for c in range(r, p):
                coefficient_matrix[r, c] = np.ascontiguousarray(lags[c - r].T)
        right_hand_side = np.vstack([lags[r + 1] for r in range(p)])
        covariance_function = lags
        model_order = p

        coefficient_matrix.cholesky()
        rhs = _dev(right_hand_side)
        x1 = coefficient_matrix.solve_triangular(rhs, transpose=True)
        x2 = coefficient_matrix.solve_triangular(x1)
        Q = _dev(covariance_function[0])
        engine.gemm(x2, rhs, transa=True, alpha=-1.0, beta=1.0, out=Q)
        return AutoregressiveModel(np.split(engine.to_host(x2).T, model_order, axis=1), engine.to_host(Q))

    @staticmethod
    def from_sample(sample, order):
        """grates/lstsq.py:170-190 (as upstream: every lag uses the zero-lag product)"""
        s = _dev(sample)
        product = engine.to_host(engine.gemm(s, s, transa=True))
        covariance_function = [product / (sample.shape[0] - k) for k in range(order + 1)]
        return AutoregressiveModel.from_covariance_function(covariance_function)

    def __compute_normals(self):
        """Normal equations of the pseudo-observations of the model (grates/lstsq.py:192-209): W = chol(Q) (upper),
        observation blocks W^-T B_k (k descending) and -W^-T, normal blocks as their products."""
        W = engine.potrf(_dev(self.__covariance_matrix))
        Winv = engine.trtri(W)
        observation_equations = [engine.gemm(Winv, _dev(B), transa=True) for B in self.__coefficients[::-1]]
        minus = _zeros(Winv.shape)
        engine.axpby(-1.0, Winv.t().contiguous(), 0.0, minus)
        observation_equations.append(minus)

        count = self.order + 1
        bounds = list(range(0, (count + 1) * self.dimension, self.dimension))
        self.__normal_equation = BlockMatrix(bounds, bounds)
        for r, left in enumerate(observation_equations):
            for c in range(r, count):
                self.__normal_equation._set_device(r, c, engine.gemm(left, observation_equations[c], transa=True))

    def normal_equation_block(self, row, column):
        """normal-equation block (row, column) as ndarray (grates/lstsq.py:211-230)"""
        return engine.to_host(self._normal_equation_block_device(row, column))

    def _normal_equation_block_device(self, row, column):
        if self.__normal_equation is None:
            self.__compute_normals()
        return self.__normal_equation.device_block(row, column)

    def to_transformed_coefficients(self):
        """grates/lstsq.py:232-247"""
        W_inv = engine.trtri(engine.potrf(_dev(self.__covariance_matrix)))
        transformed = [engine.to_host(engine.gemm(W_inv, _dev(B), alpha=-1.0)) for B in self.__coefficients[::-1]]
        transformed.append(engine.to_host(W_inv))
        return np.hstack(transformed)


class AutoregressiveModelSequence:
    """Sequence of VAR models of increasing order, starting from order 0 (grates/lstsq.py:250-411)."""

    def __init__(self, armodels):
        self.__armodels = armodels

    @staticmethod
    def from_covariance_function(covariance_function):
        return AutoregressiveModelSequence([AutoregressiveModel.from_covariance_function(covariance_function[0:k + 1])
                                            for k in range(len(covariance_function))])

    @staticmethod
    def from_sample(sample, maximum_order):
        return AutoregressiveModelSequence([AutoregressiveModel.from_sample(sample, order) for order in range(maximum_order + 1)])

    @property
    def maximum_order(self):
        return self.__armodels[-1].order

    @property
    def dimension(self):
        return self.__armodels[-1].dimension

    def __normals_block(self, epoch_count, row, column):
        """Block (row, column), row <= column, of the constraint normals of `epoch_count` epochs (grates/lstsq.py:333-362).
        The reference scans all epoch_count - p window positions; only those with column - p <= index <= row contribute."""
        N = _zeros((self.dimension, self.dimension))
        p = self.maximum_order
        for index in range(max(0, column - p), min(row, epoch_count - p - 1) + 1):
            engine.axpby(1.0, self.__armodels[-1]._normal_equation_block_device(row - index, column - index), 1.0, N)
        for order in range(p):
            if row <= order and column <= order:
                engine.axpby(1.0, self.__armodels[order]._normal_equation_block_device(row, column), 1.0, N)
        return N

    def normal_equations(self, epoch_count):
        """Block-banded inverse covariance matrix of `epoch_count` epochs with zero right-hand side (grates/lstsq.py:364-392).
        Interior blocks are equal sums; they are formed once and copied."""
        parameter_count = epoch_count * self.dimension
        block_index = np.arange(0, parameter_count + self.dimension, self.dimension, dtype=int)
        normals_matrix = BlockMatrix(block_index, block_index)
        right_hand_side = np.zeros((parameter_count, 1))
        p = self.maximum_order
        interior = {}
        for row in range(epoch_count):
            for column in range(row, min(epoch_count, row + p + 1)):
                if p <= row <= epoch_count - p - 1:          # full window range, no start-up models: depends on the lag only
                    lag = column - row
                    if lag not in interior:
                        interior[lag] = self.__normals_block(epoch_count, row, column)
                    normals_matrix._set_device(row, column, interior[lag].clone())
                else:
                    normals_matrix._set_device(row, column, self.__normals_block(epoch_count, row, column))
        return NormalEquations(normals_matrix, right_hand_side, 0.0, parameter_count)

    def covariance_function(self, maximum_lag):
        """grates/lstsq.py:394-411"""
        epochs = max(maximum_lag, self.maximum_order) + 1
        system = self.normal_equations(epochs)
        system.compute_covariance(sparse=False)                # full inverse: block (0, k) is the covariance of lag k
        return [system.matrix[0, lag] for lag in range(maximum_lag + 1)]


class BlockMatrix:
    """
    Sparse rectangular block matrix with device-resident blocks (grates/lstsq.py:414-912).  `matrix[i, j]` returns a host
    copy of a block (None when the block is empty) and `matrix[i, j] = array` stores a device copy; `device_block(i, j)`
    gives the tensor itself.
    """

    def __init__(self, row_index, column_index):
        self.shape = (len(row_index) - 1, len(column_index) - 1)
        self.__row_index = row_index
        self.__column_index = column_index
        self.__data = {}
        self.__inverse_factor = {}      # diagonal index -> inverse of the upper triangular factor block
        self._inverse_in_place = False  # the factorisation keeps U_ii^-1 in the diagonal blocks instead of U_ii (distributed chains)
        self._holds_factor_inverses = False   # ... and has done so: the diagonal blocks currently hold U_ii^-1

    def copy(self):
        """Deep copy of BlockMatrix"""
        output = BlockMatrix(self.__row_index, self.__column_index)
        for key, block in self.__data.items():
            output.__data[key] = block.clone()
        output._inverse_in_place = self._inverse_in_place
        output._holds_factor_inverses = self._holds_factor_inverses
        return output

    def __require_plain_factor(self, operation):
        """A chain factored with `_inverse_in_place` keeps U_ii^-1 where the reference's class keeps U_ii: only the solves and the
        sparse inverse understand that state."""
        if self._holds_factor_inverses:
            raise ValueError('{0}: the diagonal blocks hold the inverses of the factor blocks (in-place factorisation of '
                             'grates_amd.distributed); only solve_triangular and sparse_inverse are defined in this state'.format(operation))

    @staticmethod
    def compute_block_index(array_shape, block_size):
        """grates/lstsq.py:437-463"""
        def bounds(extent):
            # 0, block_size, 2 block_size, ..., extent (the last block may be smaller)
            return np.append(np.arange(0, extent, block_size), extent).astype(int) if extent > 0 else np.array([0])
        return bounds(array_shape[0]), bounds(array_shape[1])

    @staticmethod
    def from_array(array, row_index, column_index):
        """Block matrix from a 2D ndarray; blocks without a non-zero entry stay empty (grates/lstsq.py:466-497)."""
        if not isinstance(array, np.ndarray) or array.ndim != 2:
            raise ValueError('from_array expects a two-dimensional numpy.ndarray')
        for axis, index in enumerate((row_index, column_index)):
            if index[-1] != array.shape[axis]:
                raise ValueError('block index of axis {0} ends at {1}, the array has {2} entries there'.format(axis, index[-1], array.shape[axis]))
        block_matrix = BlockMatrix(row_index, column_index)
        for row in range(len(row_index) - 1):
            for column in range(len(column_index) - 1):
                block = array[row_index[row]:row_index[row + 1], column_index[column]:column_index[column + 1]]
                if np.count_nonzero(block):
                    block_matrix[row, column] = block
        return block_matrix

    def to_array(self):
        """2D ndarray (host) of the whole matrix (grates/lstsq.py:499-514)"""
        array = np.zeros((self.__row_index[-1], self.__column_index[-1]))
        for (row, column), block in self.__data.items():
            array[self.__row_slice(row), self.__column_slice(column)] = engine.to_host(block)
        return array

    def __check_bounds(self, i, j):
        if i > self.shape[0]:
            raise IndexError("block index {0} is out of bounds for axis 0 with size {1}".format(i, self.shape[0]))
        if j > self.shape[1]:
            raise IndexError("block index {0} is out of bounds for axis 1 with size {1}".format(j, self.shape[1]))

    def __block_shape(self, i, j):
        return int(self.__row_index[i + 1] - self.__row_index[i]), int(self.__column_index[j + 1] - self.__column_index[j])

    def __row_slice(self, i):
        return slice(int(self.__row_index[i]), int(self.__row_index[i + 1]), 1)

    def __column_slice(self, i):
        return slice(int(self.__column_index[i]), int(self.__column_index[i + 1]), 1)

    def __check_item(self, i, j, item):
        if not (isinstance(item, np.ndarray) or _is_tensor(item)):
            raise ValueError('Block matrix item must be of type ' + str(np.ndarray))
        if item.ndim != 2:
            raise ValueError('Block matrix item must be a two-dimensional ' + str(np.ndarray))
        if tuple(item.shape) != self.__block_shape(i, j):
            raise ValueError('Block matrix item at position ({0:d}, {1:d}) must be of size ({2:d}, {3:d}). '
                             'Got ({4:d}, {5:d}).'.format(i, j, *self.__block_shape(i, j), item.shape[0], item.shape[1]))

    def __setitem__(self, key, value):
        if not isinstance(key, tuple) and len(key) != 2:
            raise IndexError("Indices to block matrix must be tuples of length 2")
        self.__check_bounds(key[0], key[1])
        self.__check_item(key[0], key[1], value)
        self._set_device(key[0], key[1], _dev(value))

    def _set_device(self, i, j, tensor):
        """store a device tensor as block (i, j); no copy when it is dense row-major (what csrc/blockchol.hip assumes of every block)"""
        self.__data[(int(i), int(j))] = tensor if tensor.is_contiguous() else tensor.contiguous()
        if i == j:
            self.__inverse_factor.pop(int(i), None)

    def __getitem__(self, key):
        if not isinstance(key, tuple) and len(key) != 2:
            raise IndexError("Indices to block matrix must be tuples of length 2")
        self.__check_bounds(key[0], key[1])
        block = self.__data.get((int(key[0]), int(key[1])))
        return None if block is None else engine.to_host(block)

    def device_block(self, i, j):
        """device tensor of block (i, j), or None"""
        return self.__data.get((int(i), int(j)))

    def is_nonzero(self, row, column):
        """whether block (row, column) is stored"""
        return (int(row), int(column)) in self.__data

    def __nz(self, i, j):
        return (i, j) in self.__data

    def __matmul__(self, other):
        """C = A B over the non-zero blocks (grates/lstsq.py:651-681)"""
        if not isinstance(other, BlockMatrix):
            raise ValueError('BlockMatrix @ {0} is not defined'.format(type(other).__name__))
        product = BlockMatrix(self.__row_index, other.__column_index)
        inner = range(self.shape[1])
        for i in range(product.shape[0]):
            for j in range(product.shape[1]):
                for k in (k for k in inner if self.__nz(i, k) and other.__nz(k, j)):       # ascending k: upstream summation order
                    engine.gemm(self.__data[(i, k)], other.__data[(k, j)], beta=1.0, out=product.__set_block(i, j))
        return product

    def __set_block(self, i, j):
        """zero block on first use (grates/lstsq.py:683-696)"""
        if (i, j) not in self.__data:
            self.__data[(i, j)] = _zeros(self.__block_shape(i, j))
        return self.__data[(i, j)]

    # ---- numeric kernels: one libshg call per operation (csrc/blockchol.hip walks the blocks on the device) -------------------
    def __square_bounds(self):
        if len(self.__row_index) != len(self.__column_index) or np.any(np.asarray(self.__row_index) != np.asarray(self.__column_index)):
            raise ValueError('operation needs a square block matrix with equal row and column blocks')
        return np.ascontiguousarray(self.__row_index, dtype=np.int32)

    def __block_table(self):
        """the stored upper blocks in compressed row form (engine.BlockTable)"""
        return engine.BlockTable(self.__square_bounds(), self.__data)

    def __inverse_table(self):
        """scratch matrices that hold the inverses of the diagonal factor blocks (kept until a block changes)"""
        nb = self.shape[0]
        table = np.zeros(nb, dtype=np.uint64)
        for i in range(nb):
            if i not in self.__inverse_factor:
                size = self.__block_shape(i, i)[0]
                self.__inverse_factor[i] = self.__data[(i, i)] if self._inverse_in_place else _zeros((size, size))
            table[i] = self.__inverse_factor[i].data_ptr()
        return table

    def __allocate_fill(self):
        """symbolic factorisation: eliminating block row r couples every pair of its off-diagonal blocks (r, c), (r, d), c <= d,
        so block (c, d) of the factor is non-zero as well"""
        nb = self.shape[0]
        pattern = [set() for _ in range(nb)]
        for (i, j) in self.__data:
            if j > i:
                pattern[i].add(j)
        for i in range(nb):
            if not self.__nz(i, i):
                raise np.linalg.LinAlgError('diagonal block {0} of the matrix is empty'.format(i))
        for r in range(nb):
            cols = sorted(pattern[r])
            for a, c in enumerate(cols):
                for d in cols[a:]:
                    if not self.__nz(c, d):
                        self.__data[(c, d)] = _zeros(self.__block_shape(c, d))
                    if d != c:
                        pattern[c].add(d)

    def cholesky(self):
        """
        Cholesky factorization N = W^T W in place; only the upper triangle is referenced, afterwards the matrix holds the
        upper triangular factor W (grates/lstsq.py:698-717).  Raises numpy.linalg.LinAlgError if a diagonal block is not
        positive definite.
        """
        self.__square_bounds()
        self.__allocate_fill()
        self.__inverse_factor.clear()
        pivot = engine.block_potrf(self.__block_table(), self.__inverse_table())
        self._holds_factor_inverses = self._inverse_in_place
        if pivot:
            raise np.linalg.LinAlgError('{0}-th leading minor of the array is not positive definite'.format(pivot))

    def _cholesky_rows(self, first, last):
        """Eliminate the block rows first <= r < last only (shg_block_potrf_rows): the rows from `last` on are left as the Schur
        complement.  A sequence of calls that covers all rows in ascending order equals cholesky()."""
        self.__square_bounds()
        if first == 0:
            self.__allocate_fill()
            self.__inverse_factor.clear()
        pivot = engine.block_potrf(self.__block_table(), self.__inverse_table(), first, last)
        self._holds_factor_inverses = self._inverse_in_place
        if pivot:
            raise np.linalg.LinAlgError('{0}-th leading minor of the array is not positive definite'.format(pivot))

    def _cholesky_rows_pair(self, other, first, last):
        """_cholesky_rows(first, last) of this matrix and of `other` in one pass (shg_block_potrf_rows_pair).  The two matrices
        must have the same structure in the leading block rows and columns that the rows before `last` reach (two chains that
        need not be equally long: up to block `last`; matrices with border columns: all of them)."""
        tables, inverses = [], []
        for m in (self, other):
            m.__square_bounds()
            if first == 0:
                m.__allocate_fill()
                m.__inverse_factor.clear()
            reach = max([last] + [j for (i, j) in m.__data if i < last])
            head = {k: v for k, v in m.__data.items() if k[0] <= reach and k[1] <= reach}
            tables.append(engine.BlockTable(np.ascontiguousarray(m.__row_index[:reach + 2], dtype=np.int32), head))
            inverses.append(m.__inverse_table()[:reach + 1])
        pivots = engine.block_potrf_pair(tables[0], inverses[0], tables[1], inverses[1], first, last)
        for m in (self, other):
            m._holds_factor_inverses = m._inverse_in_place
        for pivot in pivots:
            if pivot:
                raise np.linalg.LinAlgError('{0}-th leading minor of the array is not positive definite'.format(pivot))

    def _solve_rows(self, x, transpose, first, last):
        """in place on the device tensor x [n, k]: the sweep of solve_triangular over the block rows first <= r < last only
        (shg_block_solve_rows)"""
        engine.block_solve_rows(self.__block_table(), self.__inverse_table(), bool(transpose), first, last, x)
        return x

    def _sparse_inverse_rows(self, first, last):
        """sparse_inverse() for the block rows last - 1 .. first; the blocks of the later rows hold their entries of the inverse
        already (shg_block_sparse_inverse_rows).  The matrix is an ordinary (covariance) matrix afterwards."""
        engine.block_sparse_inverse_rows(self.__block_table(), self.__inverse_table(), first, last)
        self.__inverse_factor.clear()
        self._inverse_in_place = False
        self._holds_factor_inverses = False

    def __vector(self, b):
        v = _dev(b)
        return v.reshape(1, -1) if v.dim() == 1 else v

    def multiply_triangular(self, b, transpose=False):
        """v = W b or v = W^T b with the upper triangular factor (grates/lstsq.py:719-750).  As upstream, the transposed
        branch assigns instead of accumulating (lstsq.py:743)."""
        self.__require_plain_factor('multiply_triangular')
        bd = self.__vector(b)
        v = engine.block_multiply(self.__block_table(), 1 if transpose else 0, bd)
        return _like_input(v, b)

    def multiply_symmetric(self, b):
        """v = N b for a symmetric matrix of which only the upper triangle is stored (grates/lstsq.py:752-776)"""
        bd = self.__vector(b)
        return _like_input(engine.block_multiply(self.__block_table(), 2, bd), b)

    def solve_triangular(self, b, transpose=False):
        """Solve W x = b or W^T x = b with the upper triangular block factor (grates/lstsq.py:778-821)."""
        x = self.__vector(b)                                  # a copy: solved in place
        self.__ensure_factor_inverses()
        engine.block_solve(self.__block_table(), self.__inverse_table(), bool(transpose), x)
        return _like_input(x, b)

    def __ensure_factor_inverses(self):
        """inverses of the diagonal factor blocks that are not cached (a factor that was stored block by block rather than computed
        by cholesky(): grates/lstsq.py:807, 817 invert / solve with the diagonal blocks on the fly)"""
        for i in range(self.shape[0]):
            if i not in self.__inverse_factor:
                self.__inverse_factor[i] = self.__data[(i, i)] if self._holds_factor_inverses else engine.trtri(self.__data[(i, i)])

    def sparse_inverse(self):
        """
        Sparse inverse N^-1 = W^-1 W^-T on the pattern of the Cholesky factor W held by the matrix, in place
        (grates/lstsq.py:823-846).
        """
        self.__ensure_factor_inverses()
        engine.block_sparse_inverse(self.__block_table(), self.__inverse_table())
        self.__inverse_factor.clear()
        self._inverse_in_place = False
        self._holds_factor_inverses = False

    def inverse(self):
        """
        Full inverse N^-1 = W^-1 W^-T from the Cholesky factor W held by the matrix, in place, upper triangle
        (grates/lstsq.py:848-882).
        """
        self.__require_plain_factor('inverse')
        nb = self.shape[0]
        for i in range(nb):
            for j in range(i, nb):
                self.__set_block(i, j)                        # the inverse of a banded factor is dense
        self.__ensure_factor_inverses()
        engine.block_inverse(self.__block_table(), self.__inverse_table())
        self.__inverse_factor.clear()

    def _scale(self, value):
        """Scale whole matrix with a factor."""
        self.__require_plain_factor('_scale')
        for block in self.__data.values():
            engine.axpby(value, block, 0.0, block)
        self.__inverse_factor.clear()

    def _axpy(self, factor, other):
        """Perform self += factor * other."""
        self.__require_plain_factor('_axpy')
        other.__require_plain_factor('_axpy')
        for key, block in other.__data.items():
            if key in self.__data:
                engine.axpby(factor, block, 1.0, self.__data[key])
            else:
                self.__data[key] = _zeros(block.shape)
                engine.axpby(factor, block, 0.0, self.__data[key])
        self.__inverse_factor.clear()

    def diag(self):
        """Return copy of main diagonal."""
        d = np.zeros(min(self.__row_index[-1], self.__column_index[-1]))
        for idx in range(min(len(self.__row_index), len(self.__column_index)) - 1):
            if self.__nz(idx, idx):
                d[self.__row_index[idx]:self.__row_index[idx + 1]] = engine.to_host(self.__data[(idx, idx)].diagonal())
        return d


class NormalEquations:
    """Normal equations N x = n of a least-squares problem: block matrix N (upper blocks), right-hand side n [n, 1] (ndarray or
    device tensor), l^T P l and the number of observations (grates/lstsq.py:915-1059).  `status` tracks what `matrix` currently
    holds: 'normal_matrix', 'cholesky_factor' or 'covariance_matrix'."""

    def __init__(self, normal_matrix, right_hand_side, observation_square_sum, observation_count):
        self.matrix, self.right_hand_side = normal_matrix, right_hand_side
        self.observation_square_sum, self.observation_count = observation_square_sum, observation_count
        self.status = 'normal_matrix'

    def __cholesky(self):
        """factor the matrix once; a matrix that already holds covariances cannot be factored again (ValueError, as upstream)"""
        if self.status == 'covariance_matrix' or self.status not in ('normal_matrix', 'cholesky_factor'):
            raise ValueError('the matrix holds {0}: only a normal matrix can be factored'.format(self.status))
        if self.status == 'normal_matrix':
            self.matrix.cholesky()
            self.status = 'cholesky_factor'

    def solve(self, signs=None):
        """
        Solve the system; the coefficient matrix afterwards holds the upper triangular Cholesky factor.  As upstream, 100
        Monte-Carlo vectors of random signs (numpy.random.randint, global state, drawn on the host so that a seeded run
        reproduces the reference) are solved along with the right-hand side and kept in `monte_carlo_vectors`
        (grates/lstsq.py:950-968).

        signs : ndarray or device tensor [n, k] of +-1, optional (extension)
            Monte-Carlo vectors to use instead of the host draw (5e8 draws and a 5 GB upload for config 5).
        """
        self.__cholesky()
        rhs = _dev(self.right_hand_side)
        h = self.matrix.solve_triangular(rhs, transpose=True)
        if signs is None:
            xi = np.random.randint(0, 2, size=(h.shape[0], 100))
            xi[xi == 0] = -1
            signs = xi.astype(np.float64)
        torch = engine.require_gpu()
        x = self.matrix.solve_triangular(torch.cat((h, _dev(signs)), dim=1))
        if _is_tensor(self.right_hand_side):
            self.monte_carlo_vectors = x[:, 1:]
            return x[:, 0:1]
        x = engine.to_host(x)
        self.monte_carlo_vectors = x[:, 1:]
        return x[:, 0:1]

    def redundancy(self, combined_normals, variance_factor):
        """grates/lstsq.py:970-988"""
        mc = _dev(combined_normals.monte_carlo_vectors)
        Nm = self.matrix.multiply_symmetric(mc)
        estimated_trace = _dot(mc, Nm) / mc.shape[1]
        return np.asarray(self.observation_count - estimated_trace / variance_factor).squeeze()

    def residual_square_sum(self, solution):
        """grates/lstsq.py:990-1005"""
        x = _dev(solution)
        Nx = self.matrix.multiply_symmetric(x)
        rhs = _dev(self.right_hand_side)
        return np.asarray(self.observation_square_sum - 2 * _dot(rhs, x) + _dot(x, Nx)).squeeze()

    def posterior_sigma(self, solution):
        """grates/lstsq.py:1007-1024"""
        x = _dev(solution)
        Wx = self.matrix.multiply_triangular(x)
        rhs = _dev(self.right_hand_side)
        ePe = self.observation_square_sum - 2 * _dot(rhs, x) + _dot(Wx, Wx)
        return np.sqrt(ePe / (self.observation_count - x.shape[0])).squeeze()

    def compute_covariance(self, sparse=True):
        """(sparse) inverse of the coefficient matrix (grates/lstsq.py:1026-1042)"""
        self.__cholesky()
        (self.matrix.sparse_inverse if sparse else self.matrix.inverse)()
        self.status = 'covariance_matrix'

    def to_array(self):
        """grates/lstsq.py:1044-1059"""
        rhs = engine.to_host(self.right_hand_side) if _is_tensor(self.right_hand_side) else self.right_hand_side
        return self.matrix.to_array(), rhs, self.observation_square_sum, self.observation_count


class TikhonovRegularization(NormalEquations):
    """Normal equations of a Tikhonov regularization with a diagonal regularization matrix (grates/lstsq.py:1062-1088)."""

    def __init__(self, regularization_vector, block_index, right_hand_side=None):
        weights = np.asarray(regularization_vector)
        if right_hand_side is None:                            # zero bias: zero right-hand side, l^T P l = 0
            bias, weighted_square_sum = np.zeros((block_index[-1], 1)), 0
        else:                                                  # bias b with weights w: n = w * b, l^T P l = sum(w b^2)
            weighted_square_sum = np.sum(right_hand_side**2 * weights[:, np.newaxis])
            bias = right_hand_side * weights[:, np.newaxis]
        diagonal = BlockMatrix(block_index, block_index)
        for k, (lo, hi) in enumerate(zip(block_index[:-1], block_index[1:])):
            diagonal[k, k] = np.diag(weights[lo:hi])
        super().__init__(diagonal, bias, weighted_square_sum, bias.size)


def accumulate_normals(normal_equations, variance_factors):
    """Weighted sum of normal equation systems, N = sum_k N_k / s_k^2 (same for the right-hand side and l^T P l); the observation
    counts add up unweighted (grates/lstsq.py:1091-1119)."""
    # as the reference: one factor per system is read (further factors are ignored, a missing one is an IndexError), the matrix is scaled
    # by the reciprocal, right-hand side and square sum are DIVIDED by the factor (bit-equal to grates/lstsq.py:1106-1116)
    factors = [variance_factors[k] for k in range(len(normal_equations))]
    matrix = normal_equations[0].matrix.copy()
    matrix._scale(1 / factors[0])
    for part, factor in zip(normal_equations[1:], factors[1:]):
        matrix._axpy(1 / factor, part.matrix)
    sides = [engine.to_host(part.right_hand_side) if _is_tensor(part.right_hand_side) else part.right_hand_side for part in normal_equations]
    right_hand_side = sides[0].copy() / factors[0]
    square_sum = normal_equations[0].observation_square_sum / factors[0]
    for part, side, factor in zip(normal_equations[1:], sides[1:], factors[1:]):
        right_hand_side += side / factor
        square_sum += part.observation_square_sum / factor
    count = sum(part.observation_count for part in normal_equations)
    return NormalEquations(matrix, right_hand_side, square_sum, count)


def compute_variance_factors(normal_equations, combined_normals, solution, variance_factors):
    """Variance component estimates of the individual systems (grates/lstsq.py:1122-1149)."""
    return np.array([part.residual_square_sum(solution) / part.redundancy(combined_normals, factor)
                     for part, factor in zip(normal_equations, variance_factors)])
