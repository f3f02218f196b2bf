// tests/cpp/facade_find_overlap.cpp — helper of tests/test_facade_cpp.py: reads intervals and queries from a text
// file ("n q", then n lines "low high", then q lines "qlow qhigh"), inserts the intervals in file order into the
// drop-in IntervalTree, and prints, per query, what find_overlap returns in each hit-order mode:
//   "<low> <high> | <low> <high>"   (reference-compatible mode | default mode; "-" for nullopt)
// The python side compares the first column with the oracle's single descent (interval_tree.hpp:290-304) and the
// second with "the overlapping interval inserted first".
#include <binary/algorithm/all.hpp>
#include <cstdio>
#include <fstream>
#include <vector>

using namespace binary::algorithm::tree;

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::size_t n = 0, q = 0;
  in >> n >> q;
  IntervalTree<UIntIntervalNode> tree{};
  for (std::size_t i = 0; i < n; ++i) {
    std::uint32_t lo = 0, hi = 0;
    in >> lo >> hi;
    UIntInterval iv{};
    iv.low = lo;  // assigned, not constructed: low > high entries must not trip the constructor's assert
    iv.high = hi;
    tree.insert_node(UIntIntervalNode{iv});
  }
  for (std::size_t i = 0; i < q; ++i) {
    std::uint32_t lo = 0, hi = 0;
    in >> lo >> hi;
    UIntInterval qi{};
    qi.low = lo;
    qi.high = hi;
    tree.set_hit_order(HitOrder::ReferencePreorder);
    const auto a = tree.find_overlap(qi);
    tree.set_hit_order(HitOrder::Insertion);
    const auto b = tree.find_overlap(qi);
    if (a) std::printf("%u %u", a->low, a->high); else std::printf("-");
    std::printf(" | ");
    if (b) std::printf("%u %u", b->low, b->high); else std::printf("-");
    std::printf("\n");
  }
  return 0;
}
