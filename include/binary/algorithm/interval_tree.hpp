// binary/algorithm/interval_tree.hpp — drop-in for ylab-hi/BINARY's IntervalTree / IntervalNode /
// BaseInterval (reference: library/include/binary/algorithm/interval_tree.hpp:31-187), with the overlap
// queries executed on an MI355X through the C ABI of <bivx.h> (libbivx.so, hand-written HIP for gfx950).
//
// What is the same as the reference
//   - namespace binary::algorithm::tree, the concepts, BaseInterval / IntervalNode / IntervalTree, the aliases
//     IntInterval, UIntInterval, IntIntervalNode, UIntIntervalNode, every member function name and overload
//   - closed-interval predicate low <= o.high && o.low <= high (:119-121), duplicates kept (:146-155 of the
//     reference test), low > high intervals accepted (asserted only by the two-argument constructor)
//   - find_overlaps returns value copies of the stored intervals (payload included), as a SET identical to
//     the reference's result
// What differs (documented in DESIGN.md)
//   - hit ORDER: ascending insertion order by default; the reference returns RB-tree pre-order. Call
//     set_hit_order(HitOrder::ReferencePreorder) to get the reference order (host-side re-ranking).
//   - find_overlap returns the overlapping interval inserted first; the reference returns whichever one its
//     single root-to-leaf descent meets (and misses hits when q.low == 0 meets a null left child).
//   - structure introspection (root(), minimum(), successor(), search(), to_dot(), ...) works on a host-side
//     red-black tree (rb_tree.hpp) that is replayed from the insertion sequence the first time it is asked for.
//   - keys must be std::uint32_t or std::int32_t and is_overlap must not be overridden: the device evaluates
//     the default predicate. There is no CPU fallback for the query path; without a GPU construction throws.
//   - new: find_overlaps_batch(...) answers many queries in one device pass (that is the fast path).
#ifndef BINARY_AMD_ALGORITHM_INTERVAL_TREE_HPP_
#define BINARY_AMD_ALGORITHM_INTERVAL_TREE_HPP_

#include <bivx.h>

#include <algorithm>
#include <binary/algorithm/rb_tree.hpp>
#include <binary/concepts.hpp>
#include <cassert>
#include <concepts>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <ostream>
#include <ranges>
#include <span>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace binary::algorithm::tree {

  template <typename Interval>
  concept IntervalConcept = std::semiregular<Interval> && std::movable<Interval> && requires(Interval const &i) {
    requires std::same_as<decltype(i.low), decltype(i.high)>;
    { i.low <= i.high } -> std::convertible_to<bool>;
    typename Interval::key_type;
  };

  template <typename Node>
  concept IntervalNodeConcept = NodeConcept<Node> && requires(Node &n) {
    typename Node::interval_type;
    n.max;
    n.interval;
  };

  /// Tree node as the reference exposes it (interval_tree.hpp:51-103): interval, max, key, colour, links.
  template <IntervalConcept Interval> class IntervalNode {
  public:
    using interval_type = Interval;
    using key_type = typename Interval::key_type;
    using pointer = std::unique_ptr<IntervalNode>;
    using reference_pointer = pointer &;
    using raw_pointer = IntervalNode *;

    constexpr IntervalNode() = default;
    IntervalNode(IntervalNode const &) = delete;
    IntervalNode &operator=(IntervalNode const &) = delete;
    IntervalNode(IntervalNode &&) noexcept = default;
    IntervalNode &operator=(IntervalNode &&) noexcept = default;

    template <typename... Arg>
      requires std::constructible_from<Interval, Arg...>
    explicit constexpr IntervalNode(Arg &&...args)
        : interval{std::forward<Arg>(args)...}, max{interval.high}, key{interval.low} {}
    explicit constexpr IntervalNode(Interval const &i) : interval{i}, max{i.high}, key{i.low} {}
    explicit constexpr IntervalNode(Interval &&i) : interval{std::move(i)}, max{interval.high}, key{interval.low} {}

    void copy_key(const raw_pointer other) noexcept {
      key = other->key;
      max = other->max;
      interval = other->interval;
    }
    friend std::ostream &operator<<(std::ostream &os, IntervalNode const &n) {
      return os << "IntervalNode: " << n.interval << " max: " << n.max << " key: " << n.key;
    }

    void set_color(Color c) { color_ = c; }
    [[nodiscard]] auto is_black() const -> bool { return color_ == Color::Black; }
    [[nodiscard]] auto is_red() const -> bool { return color_ == Color::Red; }
    [[nodiscard]] auto leftr() const -> raw_pointer { return left.get(); }
    [[nodiscard]] auto rightr() const -> raw_pointer { return right.get(); }

    Interval interval{};
    key_type max{};
    key_type key{};
    Color color_{Color::Black};
    pointer left{nullptr};
    pointer right{nullptr};
    raw_pointer parent{nullptr};
  };

  /// Closed interval [low, high] (reference :105-132).
  template <KeyConcept KeyType = std::uint32_t> class BaseInterval {
  public:
    using key_type = std::remove_cv_t<KeyType>;

    constexpr BaseInterval() = default;
    constexpr BaseInterval(BaseInterval const &) = default;
    constexpr BaseInterval &operator=(BaseInterval const &) = default;
    constexpr BaseInterval(BaseInterval &&) noexcept = default;
    constexpr BaseInterval &operator=(BaseInterval &&) noexcept = default;
    constexpr BaseInterval(key_type low_, key_type high_) : low{low_}, high{high_} { assert(low <= high); }
    virtual ~BaseInterval() = default;

    [[nodiscard]] virtual auto is_overlap(BaseInterval const &other) const -> bool {
      return low <= other.high && other.low <= high;
    }
    friend std::ostream &operator<<(std::ostream &os, BaseInterval const &i) {
      return os << "BaseInterval: " << i.low << "-" << i.high;
    }

    key_type low{};
    key_type high{};
  };

  using IntInterval = BaseInterval<std::int32_t>;
  using UIntInterval = BaseInterval<std::uint32_t>;
  using IntIntervalNode = IntervalNode<IntInterval>;
  using UIntIntervalNode = IntervalNode<UIntInterval>;

  /// Order of the intervals inside one find_overlaps result.
  enum class HitOrder {
    Insertion,          ///< ascending insertion index (default; straight from the device)
    ReferencePreorder,  ///< the reference's RB-tree pre-order (node, left, right), re-ranked on the host
  };

  /// Result of find_overlaps_batch: query i's hits are ids[offsets[i] .. offsets[i+1]), insertion indices.
  struct OverlapBatch {
    std::vector<std::uint64_t> offsets;
    std::vector<std::uint32_t> ids;
    [[nodiscard]] auto count(std::size_t i) const -> std::size_t { return static_cast<std::size_t>(offsets[i + 1] - offsets[i]); }
    [[nodiscard]] auto hits(std::size_t i) const -> std::span<const std::uint32_t> {
      return {ids.data() + offsets[i], count(i)};
    }
  };

  namespace detail {
    struct MaxAugment {  // what the reference does in insert_node_impl :240 and the rotations :206-228
      template <typename N> static void on_descend(N *cur, N *fresh) { cur->max = std::max(cur->max, fresh->max); }
      template <typename N> static void after_rotate(N *down, N *up) {
        up->max = std::max(up->max, down->max);
        auto m = down->interval.high;
        if (down->left != nullptr) m = std::max(m, down->left->max);
        if (down->right != nullptr) m = std::max(m, down->right->max);
        down->max = m;
      }
    };

    inline void check(int rc, const char *what) {
      if (rc != 0) throw std::runtime_error(std::string(what) + ": " + bivx_last_error());
    }

    // order-preserving map of a signed or unsigned 32-bit key onto uint32
    template <typename K> constexpr auto to_u32(K k) -> std::uint32_t {
      if constexpr (std::is_signed_v<K>) return static_cast<std::uint32_t>(k) ^ 0x80000000u;
      else return static_cast<std::uint32_t>(k);
    }

    struct IndexDeleter {
      void operator()(bivx_index *p) const noexcept { bivx_destroy(p); }
    };
  }  // namespace detail

  template <IntervalNodeConcept NodeType> class IntervalTree {
  public:
    using key_type = typename NodeType::key_type;
    using pointer = typename NodeType::pointer;
    using raw_pointer = typename NodeType::raw_pointer;
    using reference_pointer = typename NodeType::reference_pointer;
    using interval_type = typename NodeType::interval_type;
    // The device evaluates the closed-interval predicate on 32-bit unsigned coordinates. uint32_t / int32_t keys map onto
    // them one to one (the reference's two aliases, interval_tree.hpp:135-136); any other INTEGER key type (int64_t,
    // uint64_t, int16_t ...; the reference takes every totally ordered key, rb_tree.hpp:20-21) goes through an
    // order-preserving window: key - (smallest coordinate of the tree), which must fit 32 bits — genomic coordinates do.
    // A tree whose coordinates span more than 2^32 - 1 throws std::domain_error at its first query (no silent slow path);
    // queries beyond the window are clamped or answered empty on the spot, which is exact because no stored interval
    // lies out there. Keys that are not integers (double, strings) have no device form: a compile-time error.
    static_assert(std::integral<key_type> && !std::same_as<key_type, bool>,
                  "the MI355X backend evaluates the closed-interval predicate on integer keys (any width; the coordinates "
                  "of one tree must span less than 2^32)");
    static constexpr bool kNative32 = std::same_as<key_type, std::uint32_t> || std::same_as<key_type, std::int32_t>;

    /// device: HIP device ordinal (default 0, or the BIVX_DEVICE environment variable).
    explicit IntervalTree(int device = default_device()) {
      bivx_index *h = nullptr;
      detail::check(bivx_create(&h, device), "bivx_create");
      index_.reset(h);
    }
    /// One tree over several GPUs of the node (bivx_create_sharded): a plain tree has a single "chromosome", so the
    /// index is replicated on every device and each batch of queries is split between them.
    explicit IntervalTree(std::span<const int> devices) {
      bivx_index *h = nullptr;
      detail::check(bivx_create_sharded(&h, devices.data(), static_cast<int>(devices.size())), "bivx_create_sharded");
      index_.reset(h);
    }
    IntervalTree(IntervalTree &&) noexcept = default;
    auto operator=(IntervalTree &&) noexcept -> IntervalTree & = default;
    IntervalTree(const IntervalTree &) = delete;
    auto operator=(const IntervalTree &) -> IntervalTree & = delete;
    virtual ~IntervalTree() = default;

    // ---- build side (reference rb_tree.hpp:111-117,142-149) -------------------------------------------------
    template <std::ranges::input_range R>
      requires std::constructible_from<NodeType, std::ranges::range_value_t<R>>
    void insert_node(R &&range) {
      for (auto &&item : range) insert_node(std::forward<decltype(item)>(item));
    }
    void insert_node(pointer node) { push(std::move(node->interval)); }
    template <typename... Args>
      requires std::constructible_from<NodeType, Args...>
    void insert_node(Args &&...args) {
      NodeType n(std::forward<Args>(args)...);  // same construction path as the reference (node from args)
      push(std::move(n.interval));
    }

    [[nodiscard]] auto size() const -> std::size_t { return items_.size(); }
    [[nodiscard]] auto empty() const -> bool { return items_.empty(); }

    // ---- query side (reference interval_tree.hpp:152-168) ------------------------------------------------------
    /// HitOrder::Insertion (default): exact on existence, returns the overlapping interval inserted first (one device
    /// call). HitOrder::ReferencePreorder: the reference's own answer — the single root-to-leaf descent of
    /// interval_tree.hpp:290-304 on the replayed host tree, which returns whichever overlapping node the descent meets
    /// first; like the reference it goes left iff q.low <= max(left), with max(nullptr) == lowest().
    [[nodiscard]] auto find_overlap(interval_type const &q) const -> std::optional<interval_type> {
      if (order_ == HitOrder::ReferencePreorder) {
        Shape &s = shape();
        for (raw_pointer x = s.root(); x != nullptr;) {
          if (q.is_overlap(x->interval)) return x->interval;
          const key_type left_max = x->left != nullptr ? x->leftr()->max : std::numeric_limits<key_type>::lowest();
          x = q.low <= left_max ? x->leftr() : x->rightr();
        }
        return std::nullopt;
      }
      sync();
      std::uint32_t lo = 0, hi = 0;
      if (!map_query(q, lo, hi)) return std::nullopt;  // (beyond every stored coordinate)
      std::uint32_t first = BIVX_NO_HIT;
      detail::check(bivx_any(index_.get(), nullptr, &lo, &hi, 1, &first), "bivx_any");
      if (first == BIVX_NO_HIT) return std::nullopt;
      return items_[first];
    }
    template <typename... Args>
      requires binary::concepts::ArgsConstructible<interval_type, Args...>
    [[nodiscard]] auto find_overlap(Args &&...args) const -> std::optional<interval_type> {
      return find_overlap(interval_type{std::forward<Args>(args)...});
    }

    /// Accepts lvalues and rvalues (the reference only compiles for rvalues, interval_tree.hpp:161).
    [[nodiscard]] auto find_overlaps(interval_type const &q) const -> std::vector<interval_type> {
      const OverlapBatch b = find_overlaps_batch(std::span<const interval_type>(&q, 1));
      std::vector<std::uint32_t> ids(b.ids.begin(), b.ids.end());
      if (order_ == HitOrder::ReferencePreorder) rank_preorder(ids);
      std::vector<interval_type> out;
      out.reserve(ids.size());
      for (auto id : ids) out.push_back(items_[id]);
      return out;
    }
    template <typename... Args>
      requires binary::concepts::ArgsConstructible<interval_type, Args...>
    [[nodiscard]] auto find_overlaps(Args &&...args) const -> std::vector<interval_type> {
      return find_overlaps(interval_type{std::forward<Args>(args)...});
    }

    /// Many queries in one device pass; ids are insertion indices, ascending inside each query.
    [[nodiscard]] auto find_overlaps_batch(std::span<const interval_type> queries) const -> OverlapBatch {
      std::vector<std::uint32_t> lo(queries.size()), hi(queries.size());
      if constexpr (kNative32) {
        for (std::size_t i = 0; i < queries.size(); ++i) {
          lo[i] = detail::to_u32(queries[i].low);
          hi[i] = detail::to_u32(queries[i].high);
        }
        return find_overlaps_batch(lo, hi);
      } else {
        sync();  // (the window's base is the tree's smallest coordinate)
        // queries that lie beyond every stored coordinate have no hit and are left out of the device batch
        std::vector<std::size_t> sent;
        sent.reserve(queries.size());
        std::size_t m = 0;
        for (std::size_t i = 0; i < queries.size(); ++i)
          if (map_query(queries[i], lo[m], hi[m])) {
            sent.push_back(i);
            ++m;
          }
        lo.resize(m);
        hi.resize(m);
        OverlapBatch part = find_overlaps_batch(lo, hi);
        if (m == queries.size()) return part;
        OverlapBatch b;
        b.ids = std::move(part.ids);
        b.offsets.assign(queries.size() + 1, 0);
        for (std::size_t k = 0; k < m; ++k) b.offsets[sent[k] + 1] = part.offsets[k + 1] - part.offsets[k];
        for (std::size_t i = 0; i < queries.size(); ++i) b.offsets[i + 1] += b.offsets[i];
        return b;
      }
    }
    [[nodiscard]] auto find_overlaps_batch(std::span<const std::uint32_t> low_u32,
                                           std::span<const std::uint32_t> high_u32) const -> OverlapBatch {
      if (low_u32.size() != high_u32.size()) throw std::invalid_argument("find_overlaps_batch: size mismatch");
      sync();
      OverlapBatch b;
      const std::size_t q = low_u32.size();
      b.offsets.assign(q + 1, 0);
      std::uint32_t *ids = nullptr;
      detail::check(bivx_find_overlaps(index_.get(), nullptr, low_u32.data(), high_u32.data(), q, nullptr, 1,
                                       b.offsets.data(), &ids),
                    "bivx_find_overlaps");
      if (ids != nullptr) b.ids.assign(ids, ids + b.offsets[q]);
      bivx_free(ids);
      return b;
    }

    /// The stored interval with insertion index id (what a hit id refers to).
    [[nodiscard]] auto interval_at(std::size_t id) const -> interval_type const & { return items_.at(id); }

    void set_hit_order(HitOrder o) { order_ = o; }
    [[nodiscard]] auto hit_order() const -> HitOrder { return order_; }
    [[nodiscard]] auto device_index() const -> const bivx_index * { sync(); return index_.get(); }

    // ---- structure introspection: the reference's tree, replayed on the host on first use ----------------------
    [[nodiscard]] auto root() const -> raw_pointer { return shape().root(); }
    [[nodiscard]] auto size(raw_pointer n) const -> std::size_t { return shape().size(n); }
    [[nodiscard]] auto minimum(raw_pointer n) const -> raw_pointer { return shape().minimum(n); }
    [[nodiscard]] auto maximum(raw_pointer n) const -> raw_pointer { return shape().maximum(n); }
    [[nodiscard]] auto successor(raw_pointer n) const -> raw_pointer { return shape().successor(n); }
    [[nodiscard]] auto predecessor(raw_pointer n) const -> raw_pointer { return shape().predecessor(n); }
    [[nodiscard]] auto search(const key_type &key) const -> raw_pointer { return shape().search(key); }
    virtual void inorder_walk(raw_pointer n, int indent) const { shape().inorder_walk(n, indent); }
    void to_dot(std::string_view filename) const { shape().to_dot(filename); }

  private:
    class Shape : public RbTree<NodeType, detail::MaxAugment> {
      void dot_node(std::ofstream &out, raw_pointer n) const override {
        out << n->key << " [label=\"" << n->key << "-" << n->interval.high << "-" << n->max
            << "\", color=" << (n->is_black() ? "black" : "red") << ", style=bold];\n";
      }
    };

    static auto default_device() -> int {
      const char *e = std::getenv("BIVX_DEVICE");
      return e ? std::atoi(e) : 0;
    }

    // key -> device coordinate (a stored coordinate: inside the window by construction)
    static constexpr auto widen(key_type k) -> std::uint64_t {
      if constexpr (std::is_signed_v<key_type>) return static_cast<std::uint64_t>(static_cast<std::int64_t>(k)) ^ (1ull << 63);
      else return static_cast<std::uint64_t>(k);
    }
    auto map_key(key_type k) const -> std::uint32_t {
      if constexpr (kNative32) return detail::to_u32(k);
      else return static_cast<std::uint32_t>(widen(k) - widen(base_));
    }
    // a query's ends as device coordinates; false: it lies beyond every stored coordinate and has no hit. Ends outside the
    // window are clamped to it — exact, since every stored coordinate is inside (call after sync()).
    auto map_query(interval_type const &q, std::uint32_t &lo, std::uint32_t &hi) const -> bool {
      if constexpr (kNative32) {
        lo = detail::to_u32(q.low);
        hi = detail::to_u32(q.high);
        return true;
      } else {
        if (!have_window_ || q.high < base_ || q.low > top_) return false;
        lo = q.low < base_ ? 0u : map_key(q.low);
        hi = q.high > top_ ? map_key(top_) : map_key(q.high);
        return true;
      }
    }

    void push(interval_type &&i) {
      items_.push_back(std::move(i));
      shape_.reset();  // structure and pre-order ranks are stale
    }

    // uploads the intervals appended since the last query and (re)builds the device index
    void sync() const {
      // const queries on a built tree may run concurrently (the reference relies on it, mapper.cpp:130-141): the
      // lazy upload / build and the lazily replayed host tree are therefore serialised
      std::lock_guard<std::mutex> lock(*lazy_);
      if constexpr (!kNative32) {
        // the window [base_, base_ + 2^32 - 1] must hold every coordinate: a new one outside it moves the window, and
        // everything is appended again under the new base
        bool moved = false;
        for (std::size_t k = synced_; k < items_.size(); ++k)
          for (const key_type c : {items_[k].low, items_[k].high}) {
            if (!have_window_) {
              base_ = top_ = c;
              have_window_ = true;
            } else if (c < base_) {
              base_ = c;
              moved = true;
            } else if (c > top_) {
              top_ = c;
            }
          }
        if (have_window_ && widen(top_) - widen(base_) > 0xFFFFFFFFull)
          throw std::domain_error("IntervalTree: the tree's coordinates span more than 2^32 - 1, which the device index "
                                  "(32-bit coordinates relative to the smallest one) cannot hold");
        if (moved && synced_ != 0) {
          detail::check(bivx_clear(index_.get()), "bivx_clear");
          synced_ = 0;
        }
      }
      if (synced_ < items_.size()) {
        const std::size_t n = items_.size() - synced_;
        std::vector<std::uint32_t> lo(n), hi(n);
        for (std::size_t k = 0; k < n; ++k) {
          lo[k] = map_key(items_[synced_ + k].low);
          hi[k] = map_key(items_[synced_ + k].high);
        }
        detail::check(bivx_append(index_.get(), nullptr, lo.data(), hi.data(), n), "bivx_append");
        synced_ = items_.size();
      }
      if (!bivx_is_built(index_.get())) detail::check(bivx_build(index_.get()), "bivx_build");
    }

    auto shape() const -> Shape & {
      std::lock_guard<std::mutex> lock(*lazy_);
      if (!shape_) {
        shape_ = std::make_unique<Shape>();
        preorder_rank_.clear();
        node_of_.clear();
        node_of_.reserve(items_.size());
        for (auto const &it : items_) {
          auto n = std::make_unique<NodeType>(it);
          node_of_.push_back(n.get());
          shape_->insert_node(std::move(n));
        }
      }
      return *shape_;
    }

    // reorders insertion ids into the reference's pre-order (node, left, right)
    void rank_preorder(std::vector<std::uint32_t> &ids) const {
      Shape &s = shape();
      std::lock_guard<std::mutex> lock(*lazy_);
      if (preorder_rank_.size() != items_.size()) {
        preorder_rank_.assign(items_.size(), 0);
        std::vector<std::pair<raw_pointer, std::uint32_t>> where;
        where.reserve(items_.size());
        for (std::uint32_t i = 0; i < node_of_.size(); ++i) where.emplace_back(node_of_[i], i);
        std::sort(where.begin(), where.end());
        std::uint32_t rank = 0;
        std::vector<raw_pointer> stack;
        if (s.root() != nullptr) stack.push_back(s.root());
        while (!stack.empty()) {
          raw_pointer n = stack.back();
          stack.pop_back();
          auto it = std::lower_bound(where.begin(), where.end(), std::make_pair(n, std::uint32_t{0}));
          preorder_rank_[it->second] = rank++;
          if (n->right != nullptr) stack.push_back(n->rightr());
          if (n->left != nullptr) stack.push_back(n->leftr());
        }
      }
      std::sort(ids.begin(), ids.end(), [&](auto a, auto b) { return preorder_rank_[a] < preorder_rank_[b]; });
    }

    std::vector<interval_type> items_;  // insertion order; id == index
    std::unique_ptr<bivx_index, detail::IndexDeleter> index_;
    mutable std::size_t synced_{0};
    mutable key_type base_{}, top_{};   // (keys other than 32-bit ones) smallest / largest stored coordinate
    mutable bool have_window_{false};
    HitOrder order_{HitOrder::Insertion};
    std::unique_ptr<std::mutex> lazy_{std::make_unique<std::mutex>()};  // (behind a pointer: the tree stays movable)
    mutable std::unique_ptr<Shape> shape_;
    mutable std::vector<raw_pointer> node_of_;
    mutable std::vector<std::uint32_t> preorder_rank_;
  };

}  // namespace binary::algorithm::tree

#endif  // BINARY_AMD_ALGORITHM_INTERVAL_TREE_HPP_
