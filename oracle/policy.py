"""Which summation order does a linear of the hot path use?  (TEST INFRASTRUCTURE — see oracle/voxref.c.)

The numeric contract has two regimes:
  * calls of at most `exact_rows` rows (default 2; vox_ctx_set_exact_rows accepts 1..8): the canonical wave64 order
    (oracle/voxref.c: DOT, RMSNorm);
  * calls with more rows run on the matrix cores: the order is the MFMA's own arithmetic (voxref.c: vr_mfma_step8)
    composed with how the kernel that takes the call splits K.  Which kernel takes a call is a pure function of its
    shape and fusion flags — restated here from the contract in include/voxhip.h ("Summation order of a linear").

`route()` returns (dot order, norm order) for one linear; `StackPolicy` walks a decoder layer the way the engine does,
carrying the "pre-normalised input" hand-off of the split-K path.
"""
from dataclasses import dataclass
from typing import Optional

from .voxref import ORD_CANON, ORD_FULLK, ORD_MFMA4, ORD_SPLITK

PRO_COPY, PRO_RMSNORM = 0, 1
EPI_STORE, EPI_SILU, EPI_SILU_MUL = 0, 1, 2
NORM_NONE, NORM_CANON, NORM_FULLK, NORM_ROWS1024 = None, "canon", "fullk", "rows1024"


def fullk_shape_ok(B, N, K, pro, epi, exact_rows=2):
    if B <= exact_rows or B < 2 or B > 128 or K % 256:
        return False
    if N % 16 and (B > 32 or epi == EPI_SILU_MUL):
        return False
    if B > 32 and N % 128:
        return False
    ks = K // 256
    if pro == PRO_RMSNORM and epi in (EPI_STORE, EPI_SILU_MUL):
        return ks in (4, 8)
    if pro == PRO_COPY and epi == EPI_STORE:
        return ks in (4, 8, 12, 16, 24, 32)
    if pro == PRO_COPY and epi == EPI_SILU_MUL:
        return ks == 16
    return False


@dataclass
class Call:
    B: int
    N: int
    K: int
    pro: int = PRO_COPY
    epi: int = EPI_STORE
    x_out: bool = False          # the call also hands its normalised input to another consumer (codec_head -> depth)
    x_rows: bool = False         # row indirection on the input
    fixed_order: bool = False    # the engine pins the canonical kernels for this call
    splitk_ws: bool = False      # a split-K workspace is attached (decoder-stack linears)
    norm_scratch: bool = False   # scratch for a normalise-once pass is attached (decoder-stack linears)
    x_prenormed: bool = False    # the producing linear already wrote norm(x)


def fullk_prenorm_ok(c: Call, exact_rows=2):
    return (not c.fixed_order and not c.x_out and not c.x_rows and c.norm_scratch and c.pro == PRO_RMSNORM and c.K == 4096
            and c.epi in (EPI_STORE, EPI_SILU_MUL)
            and fullk_shape_ok(c.B, c.N, c.K, PRO_COPY, EPI_SILU_MUL if c.epi == EPI_SILU_MUL else EPI_STORE, exact_rows))


def is_fullk(c: Call, exact_rows=2):
    return (not c.fixed_order and not c.x_out and fullk_shape_ok(c.B, c.N, c.K, c.pro, c.epi, exact_rows)) or fullk_prenorm_ok(c, exact_rows)


def is_rows_gemm(c: Call, exact_rows=2):
    return (not is_fullk(c, exact_rows) and c.B >= 17 and not c.fixed_order and c.K % 32 == 0 and c.splitk_ws and not c.x_out
            and (c.pro == PRO_COPY or (c.pro == PRO_RMSNORM and (c.x_prenormed or (c.norm_scratch and not c.x_rows)))))


def route(c: Call, exact_rows=2):
    """-> (dot order, norm order of the prologue or None).  norm order NORM_ROWS1024 never comes from here: it is the
    order of the PREVIOUS linear's fused post-norm (x_prenormed), which the caller tracks."""
    norm = NORM_CANON if c.pro == PRO_RMSNORM else NORM_NONE
    if c.B <= exact_rows or c.fixed_order:
        return ORD_CANON, norm
    if fullk_prenorm_ok(c, exact_rows):
        return ORD_FULLK, NORM_CANON                    # normalised once by the standalone kernel, then the copy-prologue GEMM
    if is_fullk(c, exact_rows) and (c.pro == PRO_RMSNORM or c.epi == EPI_STORE):
        # (a copy-prologue SiLU*up call is full-K eligible only as the second half of the K = 4096 normalise-once pair above)
        return ORD_FULLK, (NORM_FULLK if c.pro == PRO_RMSNORM else NORM_NONE)
    if is_rows_gemm(c, exact_rows):
        return ORD_SPLITK, norm
    if c.K % 32 == 0:
        return ORD_MFMA4, norm
    return ORD_CANON, norm


@dataclass
class Policy:
    """exact_rows: rows up to which the canonical wave64 kernels are used (vox_ctx_set_exact_rows; 2 by default).
    exact_rows=None: everything canonical at any row count (no kernel does that above 8 rows; reference-fixture comparisons)."""
    exact_rows: Optional[int] = 2

    def route(self, c: Call):
        if self.exact_rows is None:
            return ORD_CANON, (NORM_CANON if c.pro == PRO_RMSNORM else NORM_NONE)
        return route(c, self.exact_rows)

    def rows_gemm(self, c: Call):
        return self.exact_rows is not None and c.B > self.exact_rows and is_rows_gemm(c, self.exact_rows)
