/* oracle/nn_host.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C fp32 restatement of the reference's leaf evaluator, used (a) as a second checker next to
 * oracle/nn_oracle.py and (b) as bench.py's `cpu_baseline` for leaf-evals/s ("port": this repo's
 * restatement on the host cores, not the Oak binary -- Eigen is absent from the reference checkout).
 *   .battle.net reader                cpp/include/nn/affine.h:35-70, nn/battle/network.h:52-70, search.cc:127-131
 *   Encode::Battle::Pokemon::write    cpp/include/encode/battle/battle.h:16-214 (sparse form)
 *   Encode::Battle::ActivePokemon     battle.h:229-551
 *   EmbeddingNet::propagate (sparse)  cpp/include/nn/ffn.h:47-51, affine.h:87-103 (column axpy, then dense)
 *   write_battle_embedding            nn/battle/network.h:131-175
 *   MainNet::propagate (value path)   nn/battle/main-net.h:57-64; sigmoid network.h:14,75
 * The reference keeps per-search embedding caches (nn/battle/cache.h); for a batch of unrelated battles each
 * evaluated once they never hit, so this port computes every embedding directly -- like the GPU path.
 * Pinned the same way as nn_oracle.py: tests/test_nn_oracle.py compares both on the torch-mirror goldens. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t in, out; float *b, *w, *wt; } affine; /* w[out][in] as in the file; wt[in][out] for the sparse layer */
typedef struct {
  int activation; /* 1 relu, 2 clamp */
  affine L[12];   /* p0 p1 a0 a1 fc0 fc1 v2 v3 q1a q1b q2a q2b */
  uint32_t pod, aod, side_dim;
} nn_net;

static float act1(const nn_net *n, float x) {
  x = x > 0.0f ? x : 0.0f;
  return n->activation == 2 && x > 1.0f ? 1.0f : x;
}

void oracle_nn_free(nn_net *n) {
  if (!n) return;
  for (int i = 0; i < 12; ++i) { free(n->L[i].b); free(n->L[i].w); free(n->L[i].wt); }
  free(n);
}

nn_net *oracle_nn_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return 0;
  nn_net *n = (nn_net *)calloc(1, sizeof *n);
  uint8_t header[8];
  if (fread(header, 1, 8, f) != 8) goto bad;
  n->activation = header[0] + 1;
  for (int i = 0; i < 12; ++i) {
    affine *a = &n->L[i];
    uint32_t d[2];
    if (fread(d, 4, 2, f) != 2 || d[0] == 0 || d[1] == 0 || d[0] > 65536 || d[1] > 65536) goto bad;
    a->in = d[0]; a->out = d[1];
    a->b = (float *)malloc(4 * (size_t)a->out);
    a->w = (float *)malloc(4 * (size_t)a->out * a->in);
    if (fread(a->b, 4, a->out, f) != a->out || fread(a->w, 4, (size_t)a->out * a->in, f) != (size_t)a->out * a->in) goto bad;
    a->wt = (float *)malloc(4 * (size_t)a->out * a->in);
    for (uint32_t o = 0; o < a->out; ++o)
      for (uint32_t k = 0; k < a->in; ++k) a->wt[(size_t)k * a->out + o] = a->w[(size_t)o * a->in + k];
  }
  if (fgetc(f) != EOF) goto bad; /* must hit EOF exactly, network.h:60-63 */
  fclose(f);
  n->pod = n->L[1].out; n->aod = n->L[3].out;
  n->side_dim = (1 + n->aod) + 5 * (1 + n->pod);
  if (n->L[0].in != 198 || n->L[2].in != 427 || n->L[4].in != 2 * n->side_dim || n->L[7].out != 1) { oracle_nn_free(n); return 0; }
  return n;
bad:
  fclose(f);
  oracle_nn_free(n);
  return 0;
}

/* y = act(W x + b): row dot products, 8-wide partial sums (what Eigen's vectorised GEMV does) */
typedef float v8 __attribute__((vector_size(32), aligned(4)));
static void dense(const nn_net *n, const affine *a, const float *x, float *y, int activate) {
  for (uint32_t o = 0; o < a->out; ++o) {
    const float *w = a->w + (size_t)o * a->in;
    v8 acc0 = {0}, acc1 = {0};
    uint32_t k = 0;
    for (; k + 16 <= a->in; k += 16) {
      acc0 += *(const v8 *)(w + k) * *(const v8 *)(x + k);
      acc1 += *(const v8 *)(w + k + 8) * *(const v8 *)(x + k + 8);
    }
    acc0 += acc1;
    float s = ((acc0[0] + acc0[4]) + (acc0[2] + acc0[6])) + ((acc0[1] + acc0[5]) + (acc0[3] + acc0[7]));
    for (; k < a->in; ++k) s += w[k] * x[k];
    s += a->b[o];
    y[o] = activate ? act1(n, s) : s;
  }
}

/* sparse first layer + dense second layer (affine.h:87-103, ffn.h:47-51) */
static void embed(const nn_net *n, const affine *l0, const affine *l1, const uint16_t *idx, const float *val, int nnz, float *out) {
  float h[128];
  memcpy(h, l0->b, 4 * (size_t)l0->out);
  for (int k = 0; k < nnz; ++k) {
    const float *col = l0->wt + (size_t)idx[k] * l0->out;
    const float v = val[k];
    for (uint32_t o = 0; o < l0->out; ++o) h[o] += col[o] * v;
  }
  for (uint32_t o = 0; o < l0->out; ++o) h[o] = act1(n, h[o]);
  dense(n, l1, h, out, 1);
}

static uint32_t u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

static int status_index(uint32_t status, uint32_t sleeps) { /* battle.h:103-123 */
  if (!(status & 7)) return __builtin_ctz(status) - 3;
  if (!(status & 0x80)) return 3 + (int)sleeps;
  return 14 - (int)(status & 7);
}

/* Encode::Battle::Pokemon::write, sparse (battle.h:208-214) */
static int encode_pokemon(const uint8_t *pk, uint32_t sleep, int offset, uint16_t *idx, float *val) {
  int n = 0;
  val[n] = (float)u16(pk) / 703.0f; idx[n++] = (uint16_t)offset;
  for (int i = 1; i < 5; ++i) { val[n] = (float)u16(pk + 2 * i) / 999.0f; idx[n++] = (uint16_t)(offset + i); }
  offset += 5;
  for (int m = 0; m < 4; ++m) {
    const uint32_t mid = pk[10 + 2 * m], pp = pk[11 + 2 * m];
    if (mid != 165 && mid != 0 && pp) { idx[n] = (uint16_t)(offset + mid - 1); val[n++] = 1.0f; }
  }
  offset += 164;
  const uint32_t status = pk[20];
  if (status) { idx[n] = (uint16_t)(offset + status_index(status, sleep)); val[n++] = 1.0f; }
  offset += 14;
  const uint32_t t1 = pk[22] & 15, t2 = pk[22] >> 4;
  idx[n] = (uint16_t)(offset + t1); val[n++] = 1.0f;
  if (t2 != t1) { idx[n] = (uint16_t)(offset + t2); val[n++] = 1.0f; }
  return n;
}

static const float BOOST_NUM[13] = {25, 28, 33, 40, 50, 66, 1, 15, 2, 25, 3, 35, 4};
static const float BOOST_DEN[13] = {100, 100, 100, 100, 100, 100, 1, 10, 1, 10, 1, 10, 1};

/* Encode::Battle::ActivePokemon::write, sparse (battle.h:544-551): Active then Pokemon */
static int encode_active_pokemon(const uint8_t *pk, const uint8_t *act, uint32_t dur, uint16_t *idx, float *val) {
  int n = 0, off = 0;
  val[n] = (float)u16(act) / 703.0f; idx[n++] = 0;
  for (int i = 1; i < 5; ++i) { val[n] = (float)u16(act + 2 * i) / 999.0f; idx[n++] = (uint16_t)i; }
  off = 5;
  const uint32_t t1 = act[11] & 15, t2 = act[11] >> 4;
  idx[n] = (uint16_t)(off + t1); val[n++] = 1.0f;
  if (t2 != t1) { idx[n] = (uint16_t)(off + t2); val[n++] = 1.0f; }
  off += 15;
  for (int i = 0; i < 6; ++i) { /* atk def spe spc acc eva, battle.h:271-285 */
    const uint32_t nib = (act[12 + (i >> 1)] >> (4 * (i & 1))) & 15;
    const int st = (int)(nib ^ 8) - 8;
    const float mult = BOOST_NUM[st + 6] / BOOST_DEN[st + 6];
    val[n] = mult * (i < 4 ? 0.25f : (float)(1.0 / 3.0));
    idx[n++] = (uint16_t)(off + i);
  }
  off += 6;
  uint64_t vol;
  memcpy(&vol, act + 16, 8);
  static const int bits[16] = {0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};
  for (int i = 0; i < 16; ++i)
    if ((vol >> bits[i]) & 1) { idx[n] = (uint16_t)(off + i); val[n++] = 1.0f; }
  const uint32_t state = (uint32_t)(vol >> 24) & 0xFFFF, sub = (uint32_t)(vol >> 40) & 0xFF, tox = (uint32_t)(vol >> 59) & 31;
  if (state) { idx[n] = (uint16_t)(off + 16); val[n++] = (float)state / 65535.0f; }
  if (sub) { idx[n] = (uint16_t)(off + 17); val[n++] = (float)sub / 177.0f; } /* 706 / 4 + 1 */
  if (tox) { idx[n] = (uint16_t)(off + 18); val[n++] = (float)tox / 16.0f; }
  off += 19;
  for (int m = 0; m < 4; ++m) {
    const uint32_t mid = act[24 + 2 * m], pp = act[25 + 2 * m];
    if (mid != 165 && mid != 0 && pp) { idx[n] = (uint16_t)(off + mid - 1); val[n++] = 1.0f; }
  }
  off += 164;
  static const int sh[4] = {18, 21, 25, 28}, nb[4] = {3, 4, 3, 3}, dims[4] = {5, 8, 3, 4};
  for (int k = 0; k < 4; ++k) {
    const uint32_t v = (dur >> sh[k]) & ((1u << nb[k]) - 1);
    if (v) { idx[n] = (uint16_t)(off + v - 1); val[n++] = 1.0f; }
    off += dims[k];
  }
  return n + encode_pokemon(pk, dur & 7, off, idx + n, val + n);
}

/* write_battle_embedding (network.h:131-175) */
void oracle_nn_embedding(const nn_net *n, const uint8_t *battle, const uint8_t *durations, float *emb) {
  uint16_t idx[512];
  float val[512];
  memset(emb, 0, 4 * 2 * (size_t)n->side_dim);
  for (int s = 0; s < 2; ++s) {
    const uint8_t *side = battle + 184 * s;
    uint32_t dur;
    memcpy(&dur, durations + 4 * s, 4);
    float *base = emb + (size_t)s * n->side_dim;
    const int sid = (int)side[176] - 1;
    if (sid >= 0) {
      const uint8_t *stored = side + 24 * sid;
      const uint32_t hp = u16(stored + 18);
      if (hp) {
        base[0] = (float)hp / (float)u16(stored);
        const int nnz = encode_active_pokemon(stored, side + 144, dur, idx, val);
        embed(n, &n->L[2], &n->L[3], idx, val, nnz, base + 1);
      }
    }
    for (int slot = 2; slot <= 6; ++slot) {
      float *o = base + (1 + n->aod) + (size_t)(slot - 2) * (1 + n->pod);
      const uint32_t pid = side[176 + slot - 1];
      if (!pid) continue;
      const uint8_t *pk = side + 24 * (pid - 1);
      const uint32_t hp = u16(pk + 18);
      if (!hp) continue;
      o[0] = (float)hp / (float)u16(pk);
      const int nnz = encode_pokemon(pk, (dur >> (3 * (slot - 1))) & 7, 0, idx, val);
      embed(n, &n->L[0], &n->L[1], idx, val, nnz, o + 1);
    }
  }
}

/* NetworkImpl::value_inference (network.h:72-79) */
float oracle_nn_value_inference(const nn_net *n, const uint8_t *battle, const uint8_t *durations) {
  float emb[2048], h0[256], h1[256], h2[256], y;
  oracle_nn_embedding(n, battle, durations, emb);
  dense(n, &n->L[4], emb, h0, 1);
  dense(n, &n->L[5], h0, h1, 1);
  dense(n, &n->L[6], h1, h2, 1);
  dense(n, &n->L[7], h2, &y, 0);
  return 1.0f / (1.0f + expf(-y));
}

typedef struct { const nn_net *n; const uint8_t *b, *d; float *v; uint32_t lo, hi; } nn_job;
static void *nn_worker(void *arg) {
  nn_job *j = (nn_job *)arg;
  for (uint32_t i = j->lo; i < j->hi; ++i) j->v[i] = oracle_nn_value_inference(j->n, j->b + (size_t)i * 384, j->d + (size_t)i * 8);
  return 0;
}
void oracle_nn_value_inference_batch(const nn_net *n, const uint8_t *battles, const uint8_t *durations, uint32_t count,
                                     float *values, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (n->L[4].out > 256 || n->L[6].out > 256 || 2 * n->side_dim > 2048) return;
  pthread_t tid[256];
  nn_job jobs[256];
  for (int t = 0; t < threads; ++t) {
    nn_job j = {n, battles, durations, values, (uint32_t)((uint64_t)count * t / threads), (uint32_t)((uint64_t)count * (t + 1) / threads)};
    jobs[t] = j;
    if (threads == 1) { nn_worker(&jobs[0]); return; }
    pthread_create(&tid[t], 0, nn_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], 0);
}
