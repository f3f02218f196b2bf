// tests/cpp/test_facade.cpp — the reference's own IntervalTree / RbTree test cases
// (test/source/test_algorithm/test_interval_tree.cpp, test_rb_tree.cpp), re-expressed against the drop-in
// headers in include/binary/algorithm/. Same inputs, same expected values; doctest is not in this image, so
// a ten-line CHECK harness stands in. Overlap queries run on the GPU through libbivx.so.
//
//   usage: test_facade [--no-gpu]     (--no-gpu runs only the host-side structure cases)
#include <algorithm>
#include <array>
#include <binary/algorithm/all.hpp>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <random>
#include <stdexcept>
#include <thread>
#include <vector>

using namespace binary::algorithm::tree;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    ++g_checks;                                                          \
    if (!(cond)) {                                                       \
      ++g_fail;                                                          \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
    }                                                                    \
  } while (0)
#define CHECK_EQ(a, b) CHECK((a) == (b))

template <typename P> int black_height(P root) {  // test_interval_tree.cpp:18-29
  if (root == nullptr) return 0;
  int l = black_height(root->leftr());
  if (l != black_height(root->rightr())) throw std::invalid_argument("black height mismatch");
  return static_cast<int>(root->is_black()) + l;
}

static const std::array<UIntInterval, 10> kFixture{
    UIntInterval(16u, 21u), UIntInterval(8u, 9u),   UIntInterval(5u, 8u),   UIntInterval(0u, 3u),
    UIntInterval(6u, 10u),  UIntInterval(15u, 23u), UIntInterval(25u, 30u), UIntInterval(17u, 19u),
    UIntInterval(19u, 20u), UIntInterval(26u, 26u)};

static void host_side_cases() {
  {  // construct interval / node (test_interval_tree.cpp:31-72)
    UIntInterval a{};
    CHECK_EQ(a.low, 0u);
    CHECK_EQ(a.high, 0u);
    UIntInterval b{1, 2};
    CHECK_EQ(b.low, 1u);
    CHECK_EQ(b.high, 2u);
    UIntIntervalNode n1{1u, 10u};
    CHECK_EQ(n1.interval.low, 1u);
    CHECK_EQ(n1.interval.high, 10u);
    CHECK_EQ(n1.max, 10u);
    UIntInterval c{100u, 2000u};
    UIntIntervalNode n2{c};
    CHECK_EQ(n2.interval.high, 2000u);
    CHECK_EQ(n2.max, 2000u);
    UIntIntervalNode n3{{1u, 20u}};
    CHECK_EQ(n3.max, 20u);
    IntervalNode<IntInterval> n4{1, 10};
    CHECK_EQ(n4.interval.low, 1);
    CHECK_EQ(n4.max, 10);
  }
  {  // RbTree<IntNode> (test_rb_tree.cpp:128-207)
    RbTree<IntNode> t{};
    for (auto k : {1, 2, 4}) t.insert_node(std::make_unique<IntNode>(k));
    CHECK_EQ(t.size(), 3u);
    CHECK_EQ(t.root()->key, 2);
    CHECK_EQ(black_height(t.root()), 1);
    auto *root = t.root();
    CHECK_EQ(t.successor(root)->key, 4);
    CHECK_EQ(t.predecessor(root)->key, 1);
    CHECK(root->leftr()->parent == root && root->rightr()->parent == root);

    RbTree<IntNode> t5{};
    for (auto k : {1, 2, 7, 4, 10}) t5.insert_node(k);
    CHECK_EQ(t5.size(), 5u);
    CHECK_EQ(t5.root()->key, 2);
    CHECK_EQ(black_height(t5.root()), 2);

    std::array<int, 21> keys{3,  7,  10, 12, 14, 15, 16, 17, 19, 20, 29, 21, 23, 26, 28, 30, 35, 38, 39, 41, 47};
    RbTree<IntNode> t21{};
    for (auto k : keys) t21.insert_node(k);
    CHECK_EQ(t21.size(), keys.size());
    CHECK_EQ(t21.root()->key, 17);
    int counter = 42;
    auto perm = keys;
    do {
      RbTree<IntNode> tp{};
      for (auto k : perm) tp.insert_node(k);
      CHECK_EQ(tp.size(), perm.size());
      black_height(tp.root());
    } while (std::next_permutation(perm.begin(), perm.end()) && --counter > 0);

    std::mt19937 gen(12345);
    std::uniform_int_distribution<> dis(1, 100000);
    for (int rep = 0; rep < 10; ++rep) {
      std::vector<int> rk(5000);
      std::generate(rk.begin(), rk.end(), [&] { return dis(gen); });
      RbTree<IntNode> tr{};
      tr.insert_node(rk);  // range overload
      CHECK_EQ(tr.size(), rk.size());
      black_height(tr.root());
      // delete the root until empty (test_rb_tree.cpp:258-283)
      for (std::size_t i = 0; i < rk.size(); ++i) {
        tr.delete_node(tr.root());
        if (i % 512 == 0) black_height(tr.root());
      }
      CHECK(tr.empty());
    }

    // delete (test_rb_tree.cpp:210-256, 285-299)
    RbTree<IntNode> td{};
    for (auto k : {1, 2, 4}) td.insert_node(k);
    auto &left_child = td.root()->left;
    td.delete_node(left_child);
    CHECK_EQ(td.size(), 2u);
    RbTree<IntNode> te{};
    te.insert_node(keys);
    te.delete_node(te.root());
    te.delete_node(te.root());
    CHECK_EQ(te.size(), 19u);
    black_height(te.root());
    for (int i = 0; i < 19; ++i) te.delete_node(te.root());
    CHECK(te.empty());
    std::array<int, 20> k1{54942, 75803, 49212, 64167, 14933, 44543, 10072, 90303, 45511, 70641,
                           59710, 3100,  98544, 55068, 45575, 4994,  66267, 24721, 17128, 72975};
    RbTree<IntNode> tk{};
    tk.insert_node(k1);
    CHECK_EQ(tk.size(), 20u);
    CHECK_EQ(tk.search(54942)->key, 54942);
    CHECK(tk.search(1) == nullptr);
    tk.to_dot("rb_tree.dot");
    CHECK(std::filesystem::exists("rb_tree.dot"));
    std::filesystem::remove("rb_tree.dot");
    for (std::size_t i = 0; i < k1.size(); ++i) tk.delete_node(tk.root());
    CHECK(tk.empty());
  }
}

static void gpu_cases() {
  {  // one tree over several devices (bivx_create_sharded; the test box has one card: {0, 0}): same answers
    const std::array<int, 2> devs{0, 0};
    IntervalTree<UIntIntervalNode> t{std::span<const int>(devs)};
    t.insert_node(kFixture);
    CHECK_EQ(t.size(), 10u);
    CHECK_EQ(t.find_overlaps(7u, 25u).size(), 8u);
    CHECK_EQ(t.find_overlaps(15u, 25u).size(), 5u);
    auto one = t.find_overlap(22u, 25u);
    CHECK(one.has_value() && one->low == 15u && one->high == 23u);
    std::vector<UIntInterval> qs;
    for (unsigned k = 0; k < 40; ++k) qs.emplace_back(k, k + 2);
    auto b = t.find_overlaps_batch(qs);
    IntervalTree<UIntIntervalNode> ref{};
    ref.insert_node(kFixture);
    auto rb = ref.find_overlaps_batch(qs);
    CHECK(b.offsets == rb.offsets && b.ids == rb.ids);
    CHECK_EQ(t.root()->key, 16u);
  }
  {  // single insert (test_interval_tree.cpp:74-85)
    IntervalTree<UIntIntervalNode> t{};
    t.insert_node(16u, 21u);
    CHECK_EQ(t.size(), 1u);
    IntervalTree<IntIntervalNode> ti{};
    ti.insert_node(16, 21);
    CHECK_EQ(ti.size(), 1u);
    CHECK_EQ(ti.find_overlaps(-5, 16).size(), 1u);
    CHECK_EQ(ti.find_overlaps(-5, 15).size(), 0u);
  }
  {  // insert multiple nodes (:87-99)
    IntervalTree<UIntIntervalNode> t{};
    t.insert_node(kFixture);
    CHECK_EQ(t.size(), kFixture.size());
    CHECK_EQ(t.root()->key, 16u);
    black_height(t.root());
    CHECK_EQ(t.root()->max, 30u);
    CHECK_EQ(t.size(t.root()), 10u);
    CHECK_EQ(t.minimum(t.root())->key, 0u);
    CHECK_EQ(t.maximum(t.root())->key, 26u);
  }
  {  // 500 sequential inserts (:101-109)
    IntervalTree<IntIntervalNode> t{};
    for (int i = 0; i < 1000; i += 2) t.insert_node(i, i + 3);
    CHECK_EQ(t.size(), 500u);
    black_height(t.root());
    CHECK_EQ(t.find_overlaps(10, 10).size(), 2u);  // [8,11] and [10,13]
  }
  {  // find overlap(s) (:111-144)
    IntervalTree<UIntIntervalNode> t{};
    t.insert_node(kFixture);
    auto one = t.find_overlap(22u, 25u);
    CHECK(one.has_value());
    CHECK_EQ(one->low, 15u);
    CHECK_EQ(one->high, 23u);
    auto none = t.find_overlap(UIntInterval{100u, 111u});
    CHECK(!none.has_value());
    auto r1 = t.find_overlaps(UIntInterval{7u, 25u});
    CHECK_EQ(r1.size(), 8u);
    auto r2 = t.find_overlaps(15u, 25u);
    CHECK_EQ(r2.size(), 5u);
    // default order: insertion order
    const std::vector<std::pair<unsigned, unsigned>> ins{{16, 21}, {15, 23}, {25, 30}, {17, 19}, {19, 20}};
    for (std::size_t i = 0; i < r2.size(); ++i) CHECK(r2[i].low == ins[i].first && r2[i].high == ins[i].second);
    // reference order on request ([16,21] [15,23] [19,20] [17,19] [25,30])
    t.set_hit_order(HitOrder::ReferencePreorder);
    auto r3 = t.find_overlaps(15u, 25u);
    const std::vector<std::pair<unsigned, unsigned>> pre{{16, 21}, {15, 23}, {19, 20}, {17, 19}, {25, 30}};
    CHECK_EQ(r3.size(), pre.size());
    for (std::size_t i = 0; i < r3.size() && i < pre.size(); ++i)
      CHECK(r3[i].low == pre[i].first && r3[i].high == pre[i].second);
    auto r4 = t.find_overlaps(7u, 25u);
    const std::vector<std::pair<unsigned, unsigned>> pre8{{16, 21}, {8, 9},   {5, 8},   {6, 10},
                                                          {15, 23}, {19, 20}, {17, 19}, {25, 30}};
    CHECK_EQ(r4.size(), pre8.size());
    for (std::size_t i = 0; i < r4.size() && i < pre8.size(); ++i)
      CHECK(r4[i].low == pre8[i].first && r4[i].high == pre8[i].second);
    t.to_dot("interval_tree.dot");
    CHECK(std::filesystem::exists("interval_tree.dot"));
    std::filesystem::remove("interval_tree.dot");
    // lvalue query (does not compile against the reference, :161; a superset is fine)
    UIntInterval q{26u, 26u};
    CHECK_EQ(t.find_overlaps(q).size(), 2u);
  }
  {  // same interval value (:146-155)
    IntervalTree<UIntIntervalNode> t{};
    for (int i = 0; i < 4; ++i) t.insert_node(1u, 4u);
    CHECK_EQ(t.size(), 4u);
    CHECK_EQ(t.find_overlaps(2u, 5u).size(), 4u);
  }
  {  // insert after a query, batch entry point, payload-carrying interval type
    struct Tagged : UIntInterval {
      using UIntInterval::UIntInterval;
      Tagged() = default;
      Tagged(std::uint32_t l, std::uint32_t h, int t) : UIntInterval(l, h), tag(t) {}
      int tag{0};
    };
    IntervalTree<IntervalNode<Tagged>> t{};
    t.insert_node(10u, 20u, 7);
    CHECK_EQ(t.find_overlaps(15u, 15u).size(), 1u);
    t.insert_node(12u, 13u, 9);
    auto r = t.find_overlaps(13u, 18u);
    CHECK_EQ(r.size(), 2u);
    CHECK(r.size() == 2 && r[0].tag == 7 && r[1].tag == 9);
    std::vector<Tagged> qs{Tagged{0u, 9u, 0}, Tagged{13u, 13u, 0}, Tagged{21u, 99u, 0}, Tagged{0u, 99u, 0}};
    auto b = t.find_overlaps_batch(qs);
    CHECK_EQ(b.offsets.size(), 5u);
    CHECK(b.count(0) == 0 && b.count(1) == 2 && b.count(2) == 0 && b.count(3) == 2);
    CHECK(b.hits(1)[0] == 0 && b.hits(1)[1] == 1);
    CHECK_EQ(t.interval_at(1).tag, 9);
  }
  {  // concurrent const queries on one tree (TraMapper shares a tree across pool threads, mapper.cpp:130-141)
    IntervalTree<UIntIntervalNode> t{};
    for (std::uint32_t i = 0; i < 20000; ++i) t.insert_node(i * 7u, i * 7u + 20u);
    std::vector<std::size_t> got(8, 0);
    std::vector<std::thread> pool;
    for (int w = 0; w < 8; ++w)
      pool.emplace_back([&, w] {
        for (std::uint32_t k = 0; k < 40; ++k) got[w] += t.find_overlaps(1000u * (k + 1) + w, 1000u * (k + 1) + w + 30u).size();
        if (w == 0) got[w] += t.root() != nullptr ? 0 : 1;  // the lazy host tree under contention too
      });
    for (auto &th : pool) th.join();
    std::size_t expect0 = 0;
    for (std::uint32_t k = 0; k < 40; ++k) expect0 += t.find_overlaps(1000u * (k + 1), 1000u * (k + 1) + 30u).size();
    CHECK_EQ(got[0], expect0);
    for (int w = 1; w < 8; ++w) CHECK(got[w] >= expect0 - 40 && got[w] <= expect0 + 40);
  }
  {  // 64-bit keys (the reference takes any totally ordered key, rb_tree.hpp:20-21): coordinates far
     // beyond 2^32 through the window relative to the tree's smallest coordinate, signed and unsigned; queries that reach
     // beyond the window or lie outside it; a later insert BELOW the window's base (everything is appended again); a tree
     // whose coordinates span more than 2^32 - 1 refuses loudly
    using I64 = BaseInterval<std::int64_t>;
    using U64 = BaseInterval<std::uint64_t>;
    IntervalTree<IntervalNode<I64>> t{};
    const std::int64_t B = 1'000'000'000'000;  // 1e12
    std::vector<I64> all;
    std::uint64_t st = 99;
    auto rnd = [&](std::uint64_t n) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      return (st >> 33) % n;
    };
    for (int i = 0; i < 3000; ++i) {
      const std::int64_t lo = B + (std::int64_t)rnd(2'000'000);
      all.emplace_back(lo, lo + (std::int64_t)rnd(3000));
      t.insert_node(all.back().low, all.back().high);
    }
    auto brute = [&](I64 const &q) {
      std::vector<std::pair<std::int64_t, std::int64_t>> v;
      for (auto const &x : all)
        if (x.low <= q.high && q.low <= x.high) v.emplace_back(x.low, x.high);
      return v;
    };
    auto same = [&](I64 const &q) {
      auto got = t.find_overlaps(q.low, q.high);
      auto exp = brute(q);
      if (got.size() != exp.size()) return false;
      for (std::size_t k = 0; k < got.size(); ++k)   // (insertion order on both sides)
        if (got[k].low != exp[k].first || got[k].high != exp[k].second) return false;
      return true;
    };
    CHECK(same(I64{B + 1000, B + 5000}));
    CHECK(same(I64{B - 10, B + 100}));                 // reaches below the window: clamped
    CHECK(same(I64{-5, B + 2'500'000}));               // everything
    CHECK(same(I64{B + 1'999'000, B + 70'000'000'000}));  // reaches far beyond the window's top
    CHECK_EQ(t.find_overlaps(std::int64_t{0}, std::int64_t{B - 1}).size(), 0u);                    // wholly below
    CHECK_EQ(t.find_overlaps(std::int64_t{B + 9'000'000}, std::int64_t{B + 9'000'100}).size(), 0u);  // wholly above
    CHECK(!t.find_overlap(std::int64_t{-100}, std::int64_t{-1}).has_value());
    CHECK(t.find_overlap(std::int64_t{B}, std::int64_t{B + 2'000'000}).has_value());
    std::vector<I64> qs;
    for (int i = 0; i < 200; ++i) {
      const std::int64_t lo = B - 500'000 + (std::int64_t)rnd(3'000'000);
      qs.emplace_back(lo, lo + (std::int64_t)rnd(5000));
    }
    auto b = t.find_overlaps_batch(qs);
    CHECK_EQ(b.offsets.size(), qs.size() + 1);
    bool batch_ok = true;
    for (std::size_t i = 0; i < qs.size(); ++i) batch_ok = batch_ok && b.count(i) == brute(qs[i]).size();
    CHECK(batch_ok);
    // an insert BELOW the base moves the window
    all.emplace_back(B - 40'000, B - 39'000);
    t.insert_node(all.back().low, all.back().high);
    CHECK(same(I64{B - 50'000, B + 10}));
    CHECK(same(I64{B + 1000, B + 5000}));
    // negative coordinates, unsigned 64-bit ones
    IntervalTree<IntervalNode<I64>> neg{};
    neg.insert_node(std::int64_t{-3'000'000'000'000}, std::int64_t{-2'999'999'999'000});
    neg.insert_node(std::int64_t{-2'999'999'999'500}, std::int64_t{-2'999'999'000'000});
    CHECK_EQ(neg.find_overlaps(std::int64_t{-2'999'999'999'400}, std::int64_t{-2'999'999'999'300}).size(), 2u);
    CHECK_EQ(neg.find_overlaps(std::int64_t{-2'999'999'998'000}, std::int64_t{5}).size(), 1u);
    IntervalTree<IntervalNode<U64>> u{};
    u.insert_node(std::uint64_t{0xFFFFFFFFFFFF0000ull}, std::uint64_t{0xFFFFFFFFFFFF00FFull});
    u.insert_node(std::uint64_t{0xFFFFFFFFFFFF0080ull}, std::uint64_t{0xFFFFFFFFFFFFFFFFull});
    CHECK_EQ(u.find_overlaps(std::uint64_t{0xFFFFFFFFFFFF00F0ull}, std::uint64_t{0xFFFFFFFFFFFF0100ull}).size(), 2u);
    CHECK_EQ(u.find_overlaps(std::uint64_t{0}, std::uint64_t{0xFFFFFFFFFFFEFFFFull}).size(), 0u);
    // a span beyond 2^32 - 1: refused, loudly
    IntervalTree<IntervalNode<I64>> wide{};
    wide.insert_node(std::int64_t{0}, std::int64_t{10});
    wide.insert_node(std::int64_t{5'000'000'000}, std::int64_t{5'000'000'010});
    bool threw = false;
    try {
      (void)wide.find_overlaps(std::int64_t{0}, std::int64_t{20});
    } catch (const std::domain_error &) {
      threw = true;
    }
    CHECK(threw);
    // the host tree (structure introspection) works for 64-bit keys as for 32-bit ones
    CHECK(t.root() != nullptr && t.size(t.root()) == all.size());
  }
  {  // empty tree
    IntervalTree<UIntIntervalNode> t{};
    CHECK(t.empty());
    CHECK(!t.find_overlap(0u, 10u).has_value());
    CHECK_EQ(t.find_overlaps(0u, 0xFFFFFFFFu).size(), 0u);
    CHECK(t.root() == nullptr);
  }
}

int main(int argc, char **argv) {
  const bool no_gpu = argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0;
  try {
    host_side_cases();
    if (!no_gpu) gpu_cases();
  } catch (const std::exception &e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
