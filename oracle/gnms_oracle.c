/*
 * oracle/gnms_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-thread CPU restatement of the reference's GrooMeD-NMS hot path
 * (abhi1kumar/groomed_nms; citations are file:line under /root/reference).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to check the
 * HIP kernels against on the GPU box, where the reference itself never travels.  Nothing under
 * groomed_nms_amd/ may import, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against the
 * golden vectors in tests/golden (npz files), which tests/golden/make_golden.py produced by importing
 * the reference's own Python (lib/groomed_nms.py, lib/core.py, lib/math_3d.py, lib/nms/py_cpu_nms.py,
 * lib/nms_others.py) in the build container, and against the two known-answer vectors of
 * test/test_differentiable_nms_forward.py:127-140.
 *
 * Arithmetic: fp32 wherever the reference is fp32 (tensors are .float(), lib/groomed_nms.py:35-36),
 * compiled with -ffp-contract=off so products and sums round separately like torch's CPU kernels.
 * Group inverses use a double-precision Gauss-Jordan (the reference calls LAPACK through
 * torch.inverse, :107/:110; neither is bit-reproducible, both are compared at 1e-4).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GNMS_PRUNE_LINEAR 0
#define GNMS_PRUNE_SIGMOIDAL 1
#define GNMS_PRUNE_SOFT_NMS 2

/* ------------------------------------------------------------------------------------------- */
/* sorting: descending by value, ties broken by lower original index (a stable descending sort). */
/* torch.sort(descending=True) (lib/groomed_nms.py:41) leaves tie order unspecified; the build   */
/* defines it as stable and the parity inputs are tie-free.  NaN sorts first (torch: NaN is the  */
/* greatest value).                                                                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct { float v; int64_t i; } kv_t;

static int kv_desc(const void* pa, const void* pb) {
    const kv_t* a = (const kv_t*)pa; const kv_t* b = (const kv_t*)pb;
    int an = isnan(a->v), bn = isnan(b->v);
    if (an != bn) return an ? -1 : 1;
    if (!an) { if (a->v > b->v) return -1; if (a->v < b->v) return 1; }
    return (a->i < b->i) ? -1 : (a->i > b->i);
}

void gnms_oracle_argsort_desc(const float* v, int64_t n, int64_t* order) {
    kv_t* t = (kv_t*)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { t[i].v = v[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(kv_t), kv_desc);
    for (int64_t i = 0; i < n; ++i) order[i] = t[i].i;
    free(t);
}

/* ------------------------------------------------------------------------------------------- */
/* lib/groomed_nms.py:167-189 pruning_function (torch branch)                                    */
/* ------------------------------------------------------------------------------------------- */
static float prune_one(float x, float thr, float temp, int method) {
    if (method == GNMS_PRUNE_LINEAR) return x;                                  /* :173-174 */
    if (method == GNMS_PRUNE_SIGMOIDAL) {                                       /* :171-172 */
        float z = (x - thr) / temp;
        return 1.0f / (1.0f + expf(-z));
    }
    /* soft_nms :175-176 : 1 - exp(-(iou^2)/temperature) */
    return 1.0f - expf(-(x * x) / temp);
}

/* d prune / d iou, used by the backward restatement (autograd of the expressions above) */
static float prune_grad_one(float x, float thr, float temp, int method) {
    if (method == GNMS_PRUNE_LINEAR) return 1.0f;
    if (method == GNMS_PRUNE_SIGMOIDAL) {
        float z = (x - thr) / temp;
        float sg = 1.0f / (1.0f + expf(-z));
        return sg * (1.0f - sg) / temp;
    }
    return expf(-(x * x) / temp) * (2.0f * x / temp);
}

int gnms_oracle_prune(const float* x, int64_t count, float thr, float temp, int method, float* out) {
    if (method < 0 || method > 2) return -1;                                    /* :177-178 NotImplementedError */
    for (int64_t i = 0; i < count; ++i) out[i] = prune_one(x[i], thr, temp, method);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* lib/core.py:178-218 intersect + :480-508 iou, mode='combinations', torch branch               */
/*   out[i][j] = IoU(a_i, b_j); areas carry no +1; a zero-area pair gives 0/0 = NaN             */
/* ------------------------------------------------------------------------------------------- */
static inline float minf_(float a, float b) { return a < b ? a : b; }   /* torch.min/max on non-NaN boxes */
static inline float maxf_(float a, float b) { return a > b ? a : b; }

void gnms_oracle_iou2d(const float* a, int64_t M, const float* b, int64_t N, float* out) {
    for (int64_t i = 0; i < M; ++i) {
        const float* pa = a + 4 * i;
        float area_a = (pa[2] - pa[0]) * (pa[3] - pa[1]);                        /* :500-501 */
        for (int64_t j = 0; j < N; ++j) {
            const float* pb = b + 4 * j;
            float area_b = (pb[2] - pb[0]) * (pb[3] - pb[1]);                    /* :502-503 */
            float w = minf_(pa[2], pb[2]) - maxf_(pa[0], pb[0]);                 /* :210-211 */
            float h = minf_(pa[3], pb[3]) - maxf_(pa[1], pb[1]);
            w = w > 0.0f ? w : (w != w ? w : 0.0f);                              /* clamp(.,0) :212 (NaN passes) */
            h = h > 0.0f ? h : (h != h ? h : 0.0f);
            float inter = w * h;                                                 /* :218 */
            float uni = (area_a + area_b) - inter;                               /* :507 */
            out[i * N + j] = inter / uni;                                        /* :508 (after the permute) */
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* lib/math_3d.py:364-435 get_corners_of_cuboid (torch branch, iou_3d_convention=True)           */
/*   params = (x, y, z, w, h, l, ry) per box; corners out as N x 3 x 8                           */
/* ------------------------------------------------------------------------------------------- */
void gnms_oracle_corners(const float* params, int64_t N, float* corners) {
    static const int x_hi[8] = {0, 1, 0, 1, 0, 1, 1, 0};   /* corners[:,0,[1,3,5,6]] = l  :401 */
    static const int y_hi[8] = {0, 0, 1, 1, 0, 0, 1, 1};   /* corners[:,1,[2,3,6,7]] = h  :402 */
    static const int z_hi[8] = {0, 0, 0, 0, 1, 1, 1, 1};   /* corners[:,2,[4,5,6,7]] = w  :403 */
    for (int64_t n = 0; n < N; ++n) {
        const float* p = params + 7 * n;
        float x = p[0], y = p[1], z = p[2], w = p[3], h = p[4], l = p[5], ry = p[6];
        float c = cosf(ry), s = sinf(ry);
        float* o = corners + n * 24;
        for (int k = 0; k < 8; ++k) {
            float cx = (x_hi[k] ? l : 0.0f) - l / 2;                             /* :426 */
            float cy = (y_hi[k] ? h : 0.0f) - h / 2;                             /* :427 */
            float cz = (z_hi[k] ? w : 0.0f) - w / 2;                             /* :428 */
            /* bmm with R = [[c,0,s],[0,1,0],[-s,0,c]]  :430 */
            float rx = c * cx + 0.0f * cy + s * cz;
            float ryy = 0.0f * cx + 1.0f * cy + 0.0f * cz;
            float rz = (-s) * cx + 0.0f * cy + c * cz;
            o[0 * 8 + k] = rx + x;                                               /* :433-435 */
            o[1 * 8 + k] = ryy + y;
            o[2 * 8 + k] = rz + z;
        }
    }
}

/* per-box axis-aligned extents used by iou3d_approximate:
 *   volume from min/max over all 8 corners per axis (get_volume, lib/core.py:434-451);
 *   y extent from all 8 corners (:365-368); x and z extents from corners {2,3,6,7} (:383-388, 463-476). */
typedef struct { float vol, y0, y1, x0, x1, z0, z1; } aabb_t;

static void aabb_of(const float* c /*3x8*/, aabb_t* o) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) {
        mn[a] = c[a * 8]; mx[a] = c[a * 8];
        for (int k = 1; k < 8; ++k) { mn[a] = fminf(mn[a], c[a * 8 + k]); mx[a] = fmaxf(mx[a], c[a * 8 + k]); }
    }
    o->vol = ((mx[0] - mn[0]) * (mx[1] - mn[1])) * (mx[2] - mn[2]);               /* torch.prod over x,y,z */
    o->y0 = mn[1]; o->y1 = mx[1];
    static const int bev[4] = {2, 3, 6, 7};
    o->x0 = o->x1 = c[0 * 8 + bev[0]]; o->z0 = o->z1 = c[2 * 8 + bev[0]];
    for (int t = 1; t < 4; ++t) {
        o->x0 = fminf(o->x0, c[0 * 8 + bev[t]]); o->x1 = fmaxf(o->x1, c[0 * 8 + bev[t]]);
        o->z0 = fminf(o->z0, c[2 * 8 + bev[t]]); o->z1 = fmaxf(o->z1, c[2 * 8 + bev[t]]);
    }
}

static float relu0(float v) { return v > 0.0f ? v : (v != v ? v : 0.0f); }

/* lib/core.py:305-421 iou3d_approximate, mode="combinations"; generalized != 0 -> method="generalized".
 * Inputs are NOT mutated (the reference overwrites y with z through a view, :379-380; see SURVEY B). */
void gnms_oracle_iou3d(const float* ca, int64_t M, const float* cb, int64_t N, int generalized,
                       float* iou_bev, float* iou_3d) {
    aabb_t* A = (aabb_t*)malloc(sizeof(aabb_t) * (size_t)(M > 0 ? M : 1));
    aabb_t* B = (aabb_t*)malloc(sizeof(aabb_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < M; ++i) aabb_of(ca + 24 * i, &A[i]);
    for (int64_t j = 0; j < N; ++j) aabb_of(cb + 24 * j, &B[j]);
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) {
            const aabb_t* a = &A[i]; const aabb_t* b = &B[j];
            float vol = a->vol + b->vol;                                         /* :357 */
            float yi = relu0(fminf(a->y1, b->y1) - fmaxf(a->y0, b->y0));         /* :371-376 */
            /* BEV rectangles (x1,y1,x2,y2) = (xmin, zmin, xmax, zmax): iou + intersect :408-413 */
            float w = relu0(fminf(a->x1, b->x1) - fmaxf(a->x0, b->x0));
            float h = relu0(fminf(a->z1, b->z1) - fmaxf(a->z0, b->z0));
            float inter = w * h;
            float area_a = (a->x1 - a->x0) * (a->z1 - a->z0);
            float area_b = (b->x1 - b->x0) * (b->z1 - b->z0);
            if (iou_bev) iou_bev[i * N + j] = inter / ((area_a + area_b) - inter);
            float i3 = inter * yi;                                               /* :415 */
            float u3 = vol - i3;                                                 /* :416 */
            float r = i3 / u3;                                                   /* :417 */
            if (generalized) {                                                   /* :390-406, :418-419 */
                float xh = relu0(fmaxf(a->x1, b->x1) - fminf(a->x0, b->x0));
                float yh = relu0(fmaxf(a->y1, b->y1) - fminf(a->y0, b->y0));
                float zh = relu0(fmaxf(a->z1, b->z1) - fminf(a->z0, b->z0));
                float vh = (xh * yh) * zh;                                       /* :406 */
                r = r - ((vh - u3) / vh);
            }
            iou_3d[i * N + j] = r;
        }
    free(A); free(B);
}

/* ------------------------------------------------------------------------------------------- */
/* lib/groomed_nms.py:208-270 get_groups                                                         */
/*   iou: n x n row-major, scores: n.  Output: groups as flat index list + per-group lengths,    */
/*   indices refer to the INPUT order (return_original_indices=True, :266-268).                  */
/*   Deviations (documented in DESIGN.md): a leader whose own entry is <= threshold makes the    */
/*   reference loop forever (:247-262); here it is removed like the NaN case.  Empty groups are   */
/*   reported with length 0 (the reference appends an empty tensor, :255).                       */
/* ------------------------------------------------------------------------------------------- */
int64_t gnms_oracle_get_groups(const float* iou, const float* scores, int64_t n, float thr, int64_t group_size,
                               int64_t* flat, int64_t* lens) {
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t* remaining = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    gnms_oracle_argsort_desc(scores, n, order);                                  /* :213 */
    int64_t nrem = n, G = 0, nflat = 0;
    for (int64_t i = 0; i < n; ++i) remaining[i] = i;                            /* shrinking_array :242 (sorted space) */
    while (nrem > 0) {                                                           /* :247 */
        int64_t lead = remaining[0];
        int64_t cnt = 0, keep = 0;
        for (int64_t r = 0; r < nrem; ++r) {
            int64_t k = remaining[r];
            float v = iou[order[k] * n + order[lead]];                           /* shrinking_iou[:,0] :249-250 */
            int high = v > thr, low = v <= thr;
            if (r == 0 && low) low = 0;                                          /* deviation: see header */
            if (high && cnt < group_size + 1) { flat[nflat++] = order[k]; ++cnt; } /* :253-255 */
            if (low) remaining[keep++] = k;                                      /* :261-262 */
        }
        lens[G++] = cnt;
        nrem = keep;                                                             /* (:258-260 early break == empty) */
    }
    free(order); free(remaining);
    return G;
}

/* ------------------------------------------------------------------------------------------- */
/* dense helpers                                                                                 */
/* ------------------------------------------------------------------------------------------- */
/* inverse of a general m x m matrix, double Gauss-Jordan with partial pivoting; returns -1 if singular */
static int invert_d(double* a, double* inv, int64_t m) {
    for (int64_t i = 0; i < m; ++i) for (int64_t j = 0; j < m; ++j) inv[i * m + j] = (i == j);
    for (int64_t c = 0; c < m; ++c) {
        int64_t p = c; double best = fabs(a[c * m + c]);
        for (int64_t r = c + 1; r < m; ++r) if (fabs(a[r * m + c]) > best) { best = fabs(a[r * m + c]); p = r; }
        if (!(best > 0.0)) return -1;
        if (p != c) for (int64_t j = 0; j < m; ++j) {
            double t = a[c * m + j]; a[c * m + j] = a[p * m + j]; a[p * m + j] = t;
            t = inv[c * m + j]; inv[c * m + j] = inv[p * m + j]; inv[p * m + j] = t;
        }
        double d = a[c * m + c];
        for (int64_t j = 0; j < m; ++j) { a[c * m + j] /= d; inv[c * m + j] /= d; }
        for (int64_t r = 0; r < m; ++r) if (r != c) {
            double f = a[r * m + c];
            if (f != 0.0) for (int64_t j = 0; j < m; ++j) { a[r * m + j] -= f * a[c * m + j]; inv[r * m + j] -= f * inv[c * m + j]; }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* lib/groomed_nms.py:10-129 differentiable_nms, hard sort (sorting_method="hard"), forward and   */
/* backward.  `presorted` != 0 reproduces what the soft-sort path hands to the same code          */
/* (:42-45): scores/iou are used in INPUT order (no gather), get_groups still re-sorts (:213).    */
/*                                                                                               */
/* outputs                                                                                       */
/*   order[n]    rank -> input index (argsort desc, :41)                                         */
/*   prob[n]     the third return value (:124-129), in the order the reference returns it        */
/*   valid/invalid, *nvalid  the first two return values (:116-123)                              */
/* backward (any of the grad pointers may be NULL):                                              */
/*   grad_prob[n] = dL/dprob  ->  grad_scores[n] (input order), grad_iou[n*n] (input order)      */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    float nms_threshold, temperature, valid_box_prob_threshold;
    int pruning_method, return_sorted_prob, group_boxes, mask_group_boxes, presorted;
    int64_t group_size;
} gnms_oracle_params;

int gnms_oracle_nms(const float* scores_in, const float* iou_in, int64_t n, const gnms_oracle_params* P,
                    int64_t* order, float* prob, int64_t* valid, int64_t* invalid, int64_t* nvalid,
                    const float* grad_prob, float* grad_scores, float* grad_iou) {
    if (P->pruning_method < 0 || P->pruning_method > 2) return -1;
    size_t nn = (size_t)n * (size_t)n;
    if (n == 0) { *nvalid = 0; return 0; }
    float* s = (float*)malloc(sizeof(float) * (size_t)n);
    float* iou = (float*)malloc(sizeof(float) * nn);
    float* phi = (float*)malloc(sizeof(float) * nn);
    float* Mx = (float*)calloc(nn, sizeof(float));                                /* inversion_matrix :65 */
    float* pre = (float*)malloc(sizeof(float) * (size_t)n);
    float* r = (float*)malloc(sizeof(float) * (size_t)n);
    float* r2 = (float*)malloc(sizeof(float) * (size_t)n);
    int64_t* sorted_idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* gflat = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* glens = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t G = 0;
    int rc = 0;

    gnms_oracle_argsort_desc(scores_in, n, order);                               /* :41 */
    if (P->presorted) for (int64_t i = 0; i < n; ++i) order[i] = i;              /* caller maps through its own :41 */
    for (int64_t i = 0; i < n; ++i) {
        int64_t oi = P->presorted ? i : order[i];
        s[i] = scores_in[oi];                                                    /* :47 */
        for (int64_t j = 0; j < n; ++j) {
            int64_t oj = P->presorted ? j : order[j];
            iou[i * n + j] = iou_in[oi * n + oj];                                /* :48 */
        }
    }
    /* prune, tril, zero diagonal :71-73 */
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            phi[i * n + j] = (j < i) ? prune_one(iou[i * n + j], P->nms_threshold, P->temperature, P->pruning_method) : 0.0f;

    if (P->group_boxes) {
        G = gnms_oracle_get_groups(iou, s, n, P->nms_threshold, P->group_size, gflat, glens);   /* :85 */
        if (P->mask_group_boxes) {                                               /* :95-100 */
            /* mask[g, g[0]] = 1 ; phi *= mask  ==> keep phi[i][head] for i in g, zero elsewhere */
            uint8_t* mask = (uint8_t*)calloc(nn, 1);
            int64_t off = 0;
            for (int64_t g = 0; g < G; ++g) {
                if (glens[g] > 0) { int64_t head = gflat[off]; for (int64_t t = 0; t < glens[g]; ++t) mask[gflat[off + t] * n + head] = 1; }
                off += glens[g];
            }
            for (size_t e = 0; e < nn; ++e) if (!mask[e]) phi[e] = 0.0f;
            free(mask);
        }
        int64_t off = 0;
        for (int64_t g = 0; g < G; ++g) {                                        /* :103-108 */
            int64_t m = glens[g]; const int64_t* idx = gflat + off; off += m;
            if (m == 0) continue;                                                /* reference raises IndexError here */
            if (P->mask_group_boxes) {
                for (int64_t a = 0; a < m; ++a) for (int64_t b = 0; b < m; ++b)
                    Mx[idx[a] * n + idx[b]] = (a == b ? 1.0f : 0.0f) - phi[idx[a] * n + idx[b]];   /* :105 */
            } else {
                double* A = (double*)malloc(sizeof(double) * (size_t)(m * m));
                double* Ai = (double*)malloc(sizeof(double) * (size_t)(m * m));
                for (int64_t a = 0; a < m; ++a) for (int64_t b = 0; b < m; ++b)
                    A[a * m + b] = (a == b ? 1.0 : 0.0) + (double)phi[idx[a] * n + idx[b]];        /* :107 */
                if (invert_d(A, Ai, m) != 0) rc = -2;
                for (int64_t a = 0; a < m; ++a) for (int64_t b = 0; b < m; ++b) Mx[idx[a] * n + idx[b]] = (float)Ai[a * m + b];
                free(A); free(Ai);
            }
        }
    } else {                                                                     /* :110 inverse(I + phi) */
        /* I + phi is unit lower triangular: its inverse by forward substitution on identity columns, in double */
        double* inv = (double*)calloc(nn, sizeof(double));
        for (int64_t c = 0; c < n; ++c) {
            inv[c * n + c] = 1.0;
            for (int64_t i = c + 1; i < n; ++i) {
                double acc = 0.0;
                for (int64_t j = c; j < i; ++j) acc += (double)phi[i * n + j] * inv[j * n + c];
                inv[i * n + c] = -acc;
            }
        }
        for (size_t e = 0; e < nn; ++e) Mx[e] = (float)inv[e];
        free(inv);
    }

    /* :111 clamp(matmul(M, scores), 0, 1) -- fp32, left-to-right accumulation over the row */
    for (int64_t i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int64_t j = 0; j < n; ++j) { float m = Mx[i * n + j]; if (m != 0.0f) acc += m * s[j]; }
        pre[i] = acc;
        r2[i] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);                   /* clone :114 (NaN propagates) */
        if (acc != acc) r2[i] = acc;
        r[i] = (r2[i] < P->valid_box_prob_threshold) ? 0.0f : r2[i];             /* :115 */
    }
    gnms_oracle_argsort_desc(r, n, sorted_idx);                                  /* :117 / :121 */
    int64_t nv = 0, ni = 0;
    for (int64_t k = 0; k < n; ++k) {
        float v = r[sorted_idx[k]];
        if (v >= P->valid_box_prob_threshold) valid[nv++] = order[sorted_idx[k]];     /* :118 / :122 */
        else if (v < P->valid_box_prob_threshold) invalid[ni++] = order[sorted_idx[k]]; /* :119 / :123 */
    }
    *nvalid = nv;
    for (int64_t k = 0; k < n; ++k) {
        if (P->return_sorted_prob) prob[k] = r[sorted_idx[k]];                   /* :117 */
        else prob[k] = P->group_boxes ? r2[k] : r[k];                            /* :124-127 */
    }

    /* ------------------------------ backward (autograd of the graph above) -------------------- */
    if (grad_prob && (grad_scores || grad_iou)) {
        float* gx = (float*)calloc((size_t)n, sizeof(float));                    /* dL/d(M s), sorted space */
        for (int64_t k = 0; k < n; ++k) {
            int64_t i = P->return_sorted_prob ? sorted_idx[k] : k;
            float g = grad_prob[k];
            int thresholded = P->return_sorted_prob || !P->group_boxes;          /* which tensor was returned */
            if (thresholded && r2[i] < P->valid_box_prob_threshold) g = 0.0f;    /* in-place zeroing :115 */
            if (!(pre[i] >= 0.0f && pre[i] <= 1.0f)) g = 0.0f;                   /* clamp passes grad at the bounds */
            gx[i] = g;
        }
        float* gs = (float*)calloc((size_t)n, sizeof(float));                    /* dL/ds (sorted) = M^T gx */
        for (int64_t i = 0; i < n; ++i) if (gx[i] != 0.0f)
            for (int64_t j = 0; j < n; ++j) { float m = Mx[i * n + j]; if (m != 0.0f) gs[j] += m * gx[i]; }
        if (grad_scores) {
            for (int64_t i = 0; i < n; ++i) grad_scores[i] = 0.0f;
            for (int64_t i = 0; i < n; ++i) grad_scores[P->presorted ? i : order[i]] += gs[i];
        }
        if (grad_iou) {
            /* dL/dM = gx s^T.  masked: M = I - phi on group blocks  => dL/dphi = -dL/dM on those entries.
             * inverse: M = (I+phi_g)^-1 => dL/dphi_g = -M^T (dL/dM) M^T = -(M^T gx)(M s)^T = -gs_g * pre_g^T.
             * phi = tril(prune(iou)) with zero diagonal (and the group mask), so only j<i entries carry on. */
            for (size_t e = 0; e < nn; ++e) grad_iou[e] = 0.0f;
            float* gphi = (float*)calloc(nn, sizeof(float));
            if (P->group_boxes) {
                int64_t off = 0;
                for (int64_t g = 0; g < G; ++g) {
                    int64_t m = glens[g]; const int64_t* idx = gflat + off; off += m;
                    if (m == 0) continue;
                    if (P->mask_group_boxes) {
                        int64_t head = idx[0];
                        for (int64_t a = 0; a < m; ++a) if (idx[a] > head) gphi[idx[a] * n + head] = -(gx[idx[a]] * s[head]);
                    } else {
                        /* M is block diagonal, so the block-local M^T gx and M s are gs and pre */
                        for (int64_t a = 0; a < m; ++a) for (int64_t b = 0; b < m; ++b) if (idx[b] < idx[a])
                            gphi[idx[a] * n + idx[b]] = -(gs[idx[a]] * pre[idx[b]]);
                    }
                }
            } else {
                for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < i; ++j) gphi[i * n + j] = -(gs[i] * pre[j]);
            }
            for (int64_t i = 0; i < n; ++i) for (int64_t j = 0; j < i; ++j) {
                float gp = gphi[i * n + j];
                if (gp == 0.0f) continue;
                float d = prune_grad_one(iou[i * n + j], P->nms_threshold, P->temperature, P->pruning_method);
                int64_t oi = P->presorted ? i : order[i], oj = P->presorted ? j : order[j];
                grad_iou[oi * n + oj] += gp * d;
            }
            free(gphi);
        }
        free(gx); free(gs);
    }
    free(s); free(iou); free(phi); free(Mx); free(pre); free(r); free(r2); free(sorted_idx); free(gflat); free(glens);
    return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* lib/groomed_nms.py:131-165 soft_sort                                                          */
/*   E[i][j] = exp((-|s_j - shat_i| - rowmax_i)/T);  Z_i = sum_j E[i][j] + 1e-3                   */
/*   C[i][j] = E[i][j] / Z_j   <-- the reference divides an (n,n) tensor by an (n,) tensor (:155), */
/*   which broadcasts over the LAST axis: column j is divided by row j's sum.  Replicated as is.   */
/*   soft = C s (:158);  Cm = C iou (:163)                                                        */
/* ------------------------------------------------------------------------------------------- */
void gnms_oracle_soft_sort(const float* scores, const float* iou, int64_t n, float temperature,
                           float* C, float* soft_scores, float* soft_matrix) {
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    float* Z = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    gnms_oracle_argsort_desc(scores, n, order);                                  /* :145 */
    for (int64_t i = 0; i < n; ++i) {
        float shat = scores[order[i]];
        float mx = -INFINITY;
        for (int64_t j = 0; j < n; ++j) { float a = -fabsf(scores[j] - shat); if (a > mx) mx = a; }   /* :149-150 */
        float sum = 0.0f;
        for (int64_t j = 0; j < n; ++j) { float e = expf((-fabsf(scores[j] - shat) - mx) / temperature); C[i * n + j] = e; sum += e; }
        Z[i] = sum + 1e-3f;                                                      /* :154 */
    }
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t j = 0; j < n; ++j) C[i * n + j] = C[i * n + j] / Z[j];      /* :155 (last-axis broadcast) */
        float acc = 0.0f;
        for (int64_t j = 0; j < n; ++j) acc += C[i * n + j] * scores[j];         /* :158 */
        soft_scores[i] = acc;
    }
    if (iou && soft_matrix)
        for (int64_t i = 0; i < n; ++i) for (int64_t c = 0; c < n; ++c) {        /* :163 C @ full_matrix */
            float acc = 0.0f;
            for (int64_t j = 0; j < n; ++j) acc += C[i * n + j] * iou[j * n + c];
            soft_matrix[i * n + c] = acc;
        }
    free(order); free(Z);
}

/* ------------------------------------------------------------------------------------------- */
/* classical NMS: lib/nms/nms_kernel.cu:24-32 devIoU (+1 pixel), :61-76 strict '>' suppression,   */
/* :127-139 sequential scan.  boxes are N x boxes_dim, ALREADY sorted by score (gpu_nms.pyx:25-28) */
/* rule: 0 = GPU '>' (nms_kernel.cu:71); 1 = Cython CPU '>=' (cpu_nms.pyx:65);                    */
/*       2 = NumPy keeps '<=' i.e. suppresses '>' and drops NaN (py_cpu_nms.py:35)                 */
/* ------------------------------------------------------------------------------------------- */
static float iou_plus1(const float* a, const float* b) {
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

void gnms_oracle_classic_nms(const float* boxes, int64_t n, int64_t boxes_dim, float thresh, int rule,
                             int32_t* keep, int32_t* num_out) {
    uint8_t* removed = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    int32_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (removed[i]) continue;
        keep[k++] = (int32_t)i;
        for (int64_t j = i + 1; j < n; ++j) {
            if (removed[j]) continue;
            float v = iou_plus1(boxes + i * boxes_dim, boxes + j * boxes_dim);
            int sup = rule == 0 ? (v > thresh) : rule == 1 ? (v >= thresh) : !(v <= thresh);
            if (sup) removed[j] = 1;
        }
    }
    *num_out = k;
    free(removed);
}

/* ------------------------------------------------------------------------------------------- */
/* lib/loss/aploss.py:14-78 backpropAPLoss.forward (AP-loss, Chen et al. CVPR 2019): the consumer  */
/* of the rescored scores (lib/loss/rpn_3d.py:1117-1131), SURVEY 8-f1.                             */
/*   loss_out[0] = 1 - mean interpolated precision (:76-78); grad[n] = d loss / d logits (:69-74), */
/*   which the reference computes in forward and multiplies by grad_output in backward (:80-85).   */
/* delta is forced to 1.0 by the reference (:16) whatever the caller passes.  Row sums are taken   */
/* in double (torch.sum's vectorised fp32 order is not reproducible; compared at 1e-5).            */
/* ------------------------------------------------------------------------------------------- */
void gnms_oracle_aploss(const float* logits, const float* targets, int64_t n, float positive_label, float negative_label,
                        float* loss_out, float* grad) {
    const float delta = 1.0f;                                                    /* :16 */
    for (int64_t i = 0; i < n; ++i) grad[i] = 0.0f;                              /* :18 */
    loss_out[0] = 0.0f;                                                          /* :19 metric = zeros(1) */
    float tmax = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (targets[i] > tmax) tmax = targets[i];
    if (n == 0 || tmax <= 0.0f) return;                                          /* :26-28 */
    int64_t* fg = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t* bg = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int64_t F = 0, G = 0;
    float fmin = INFINITY;
    for (int64_t i = 0; i < n; ++i) if (targets[i] == positive_label) { fg[F++] = i; if (logits[i] < fmin) fmin = logits[i]; }   /* :30-31 */
    if (F == 0) { free(fg); free(bg); return; }   /* max(targets) > 0 without any positive_label entry: torch.min of an empty tensor raises */
    const float threshold_logit = fmin - delta;                                  /* :32 */
    for (int64_t i = 0; i < n; ++i) if (targets[i] == negative_label && logits[i] >= threshold_logit) bg[G++] = i;           /* :35 */
    float* bg_grad = (float*)calloc((size_t)(G > 0 ? G : 1), sizeof(float));     /* :37 */
    float* prec = (float*)calloc((size_t)F, sizeof(float));                      /* :44 */
    float* fgv = (float*)malloc(sizeof(float) * (size_t)F);
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)F);
    for (int64_t k = 0; k < F; ++k) fgv[k] = -logits[fg[k]];
    gnms_oracle_argsort_desc(fgv, F, order);                                     /* ascending sort of fg_logits (:47), stable */
    float max_prec = 0.0f;                                                       /* :48 */
    float* tmp2 = (float*)malloc(sizeof(float) * (size_t)(G > 0 ? G : 1));
    for (int64_t oi = 0; oi < F; ++oi) {                                         /* :50 */
        const int64_t ii = order[oi];
        const float x = logits[fg[ii]];
        double sa = 0.0, sb = 0.0;
        for (int64_t k = 0; k < F; ++k) {                                        /* :52-53 */
            float t = (logits[fg[k]] - x) / (2 * delta) + 0.5f;
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            sa += t;
        }
        for (int64_t j = 0; j < G; ++j) {                                        /* :55-56 */
            float t = (logits[bg[j]] - x) / (2 * delta) + 0.5f;
            t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
            tmp2[j] = t;
            sb += t;
        }
        const float a = (float)sa + 0.5f;                                        /* :58 */
        const float b = (float)sb;                                               /* :60 */
        const float denom = a + b;
        const float current_prec = a / denom;                                    /* :62 */
        float scale = 1.0f;
        int rescale = 0;
        if (max_prec <= current_prec) max_prec = current_prec;                   /* :63-64 */
        else { scale = (1 - max_prec) / (1 - current_prec); rescale = 1; }       /* :65-66 */
        for (int64_t j = 0; j < G; ++j) {
            float t = tmp2[j] / denom;                                           /* :61 */
            if (rescale) t *= scale;                                             /* :66 */
            bg_grad[j] += t;                                                     /* :67 */
        }
        prec[ii] = max_prec;                                                     /* :68 */
    }
    const float fnum = (float)(F > 1 ? F : 1);                                   /* :73 */
    for (int64_t j = 0; j < G; ++j) grad[bg[j]] = bg_grad[j];                    /* :70 */
    double sp = 0.0;
    for (int64_t k = 0; k < F; ++k) { grad[fg[k]] = -(1 - prec[k]); sp += prec[k]; }   /* :71 */
    for (int64_t i = 0; i < n; ++i) grad[i] /= fnum;                             /* :75 */
    loss_out[0] = 1.0f - (float)sp / fnum;                                       /* :77-78 */
    free(fg); free(bg); free(bg_grad); free(prec); free(fgv); free(order); free(tmp2);
}
