// topology.h — compile-time kinematic tree of the DeepMimic humanoid (dp_env_v3.xml:21-107).
//
// The env kernels are specialised to this tree: 13 moving bodies, one free joint + 28 hinges = 34 dofs.
// Everything derived from the parent array (dof parents, depths, the tree-sparse mass-matrix layout in
// MuJoCo's `dof_Madr` order, ancestor chains as bit masks, subtree membership) is generated here by
// constexpr code, so the solver loops below can be fully unrolled with static register indices.
// dm_model_create() verifies that the model tables it is handed describe exactly this tree.
#pragma once

namespace dmt {

constexpr int NB = 14, NV = 34, NQ = 35, NJ = 29, NG = 16, NU = 28, NOBS = 56;
constexpr int MAXPAIR = 128, MAXEFC = 64;   // one lane per constraint row ...
constexpr int MAXROWS = 63, TAU_LANE = 63;  // ... except the last, which carries the smooth force through the rows' half solve
constexpr int AOVF_COLS = MAXEFC;     // per-env memory strip for columns of A: [AOVF_COLS][64] (overflow columns; all columns during a PGS replay)
constexpr int MAXDEPTH_BODY = 4;

struct Topo {
  int body_parent[NB];
  int body_dofnum[NB];
  int body_dofadr[NB];
  int body_depth[NB];        // root = 1
  int dof_body[NV];
  int dof_parent[NV];
  int dof_depth[NV];         // number of dofs on the chain incl. itself (root x = 1)
  int madr[NV + 1];          // start of row i in the sparse M: entries (i,i), (i,parent(i)), ...
  int nM;                    // number of stored entries (310)
  short ent_i[320], ent_j[320];
  unsigned long long chain[NB];   // bit d set <=> dof d moves body b
  unsigned short subtree[NB];     // bit c set <=> body c is b or a descendant of b
  int dof_anc[NV][16];       // dof_anc[i][a] = a-th ancestor of i (a = 0 is i itself), -1 padded
};

constexpr Topo make_topo() {
  Topo t{};
  constexpr int parent[NB] = {0, 0, 1, 2, 2, 4, 2, 6, 1, 8, 9, 1, 11, 12};
  constexpr int dofnum[NB] = {0, 6, 3, 3, 3, 1, 3, 1, 3, 1, 3, 3, 1, 3};
  int adr = 0;
  for (int b = 0; b < NB; b++) {
    t.body_parent[b] = parent[b];
    t.body_dofnum[b] = dofnum[b];
    t.body_dofadr[b] = dofnum[b] ? adr : -1;
    t.body_depth[b] = b == 0 ? 0 : t.body_depth[parent[b]] + 1;
    for (int k = 0; k < dofnum[b]; k++) t.dof_body[adr + k] = b;
    adr += dofnum[b];
  }
  for (int d = 0; d < NV; d++) {
    int b = t.dof_body[d];
    if (d > t.body_dofadr[b]) t.dof_parent[d] = d - 1;
    else {
      int p = parent[b];
      t.dof_parent[d] = p > 0 ? t.body_dofadr[p] + dofnum[p] - 1 : -1;
    }
    t.dof_depth[d] = t.dof_parent[d] < 0 ? 1 : t.dof_depth[t.dof_parent[d]] + 1;
  }
  int n = 0;
  for (int i = 0; i < NV; i++) {
    t.madr[i] = n;
    int a = 0;
    for (int k = 0; k < 16; k++) t.dof_anc[i][k] = -1;
    for (int j = i; j >= 0; j = t.dof_parent[j]) {
      t.ent_i[n] = (short)i; t.ent_j[n] = (short)j; n++;
      t.dof_anc[i][a++] = j;
    }
  }
  t.madr[NV] = n; t.nM = n;
  for (int b = 0; b < NB; b++) {
    t.chain[b] = 0; t.subtree[b] = 0;
    if (b > 0 && dofnum[b] > 0)
      for (int j = t.body_dofadr[b] + dofnum[b] - 1; j >= 0; j = t.dof_parent[j]) t.chain[b] |= 1ull << j;
    for (int c = b; c < NB; c++) {
      int a = c;
      while (a > 0 && a != b) a = parent[a];
      if (a == b && !(b == 0 && c != 0)) t.subtree[b] |= (unsigned short)(1u << c);
    }
  }
  return t;
}

}  // namespace dmt
