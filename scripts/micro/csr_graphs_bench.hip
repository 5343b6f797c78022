// k_csr_graphs (csrc/k_gcn.hip) alone over G synthetic graphs of ~68 nodes / ~470 edges: time against G.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I drl_graph_exploration_amd/csrc -I include scripts/micro/csr_graphs_bench.hip -o /tmp/csrb
#include "../../drl_graph_exploration_amd/csrc/k_gcn.hip"
#include <cstdio>
#include <vector>
int main() {
  const int Gmax = 512, n = 68, m = 470;
  std::vector<int64_t> ei(2 * (size_t)Gmax * m);
  std::vector<float> ew((size_t)Gmax * m);
  std::vector<int> no(Gmax + 1), eo(Gmax + 1);
  srand(1);
  for (int g = 0; g <= Gmax; ++g) { no[g] = g * n; eo[g] = g * m; }
  const size_t E = (size_t)Gmax * m;
  for (int g = 0; g < Gmax; ++g)
    for (int j = 0; j < m; ++j) {
      int r = rand() % n, c = rand() % n;
      if (c == r) c = (r + 1) % n;
      ei[(size_t)g * m + j] = g * n + r;
      ei[E + (size_t)g * m + j] = g * n + c;
      ew[(size_t)g * m + j] = 1.0f + (rand() % 5);
    }
  int64_t *d_ei; float *d_ew, *deg, *sw, *wd, *ws_; int *d_no, *d_eo, *pd, *ed, *nd, *ps, *es, *ns;
  const int N = Gmax * n;
  hipMalloc(&d_ei, ei.size() * 8); hipMalloc(&d_ew, ew.size() * 4); hipMalloc(&d_no, (Gmax + 1) * 4); hipMalloc(&d_eo, (Gmax + 1) * 4);
  hipMalloc(&deg, N * 4); hipMalloc(&sw, N * 4); hipMalloc(&wd, E * 4); hipMalloc(&ws_, E * 4);
  hipMalloc(&pd, (N + 1) * 4); hipMalloc(&ed, (N + 1) * 4); hipMalloc(&nd, E * 4); hipMalloc(&ps, (N + 1) * 4); hipMalloc(&es, (N + 1) * 4); hipMalloc(&ns, E * 4);
  hipMemcpy(d_ei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice); hipMemcpy(d_ew, ew.data(), ew.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_no, no.data(), (Gmax + 1) * 4, hipMemcpyHostToDevice); hipMemcpy(d_eo, eo.data(), (Gmax + 1) * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int G : {1, 8, 32, 64, 128, 256, 512}) {
    for (int it = 0; it < 3; ++it)
      hipLaunchKernelGGL(k_csr_graphs, dim3(G), dim3(256), 4 * 512 * 4, 0, N, (int)E, 512, 1, d_ei, d_ew, d_no, d_eo, deg, sw, pd, ed, nd, wd, ps, es, ns, ws_, nullptr, 0, nullptr, 0);
    hipEventRecord(a, 0);
    for (int it = 0; it < 20; ++it)
      hipLaunchKernelGGL(k_csr_graphs, dim3(G), dim3(256), 4 * 512 * 4, 0, N, (int)E, 512, 1, d_ei, d_ew, d_no, d_eo, deg, sw, pd, ed, nd, wd, ps, es, ns, ws_, nullptr, 0, nullptr, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("G %4d: %.1f us per launch\n", G, ms / 20 * 1e3);
  }
  return 0;
}
