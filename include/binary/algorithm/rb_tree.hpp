// binary/algorithm/rb_tree.hpp — host-side red-black tree with the public interface of ylab-hi/BINARY's
// RbTree (reference: library/include/binary/algorithm/rb_tree.hpp:97-171), written from scratch.
//
// Role in this repository: NOT the hot path. The interval-overlap hot path (insert + find_overlaps) runs on
// the GPU behind include/bivx.h; this class exists so that code written against the reference keeps
// compiling and behaving the same where it inspects tree STRUCTURE (root(), minimum(), successor(), search(),
// delete_node(), to_dot(), colours, black heights). IntervalTree builds one lazily, only when asked.
//
// Same observable behaviour as the reference: keys equal to a node's key go to the right subtree
// (rb_tree.hpp:352-356), CLRS insert/delete fix-ups (:304-344, :430-495), nodes are owned through
// std::unique_ptr children and a raw parent pointer (:69-73).
#ifndef BINARY_AMD_ALGORITHM_RB_TREE_HPP_
#define BINARY_AMD_ALGORITHM_RB_TREE_HPP_

#include <cassert>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <memory>
#include <ranges>
#include <string>
#include <string_view>
#include <utility>

namespace binary::algorithm::tree {

  template <typename T>
  concept KeyConcept = std::totally_ordered<T> && std::default_initializable<T>;

  template <typename Node>
  concept NodeConcept = std::movable<Node> && std::default_initializable<Node> && requires(Node &n) {
    typename Node::key_type;
    typename Node::pointer;
    typename Node::reference_pointer;
    typename Node::raw_pointer;
    n.key;
    n.left;
    n.right;
    n.parent;
    n.color_;
    { n.is_black() } -> std::convertible_to<bool>;
    { n.is_red() } -> std::convertible_to<bool>;
  };

  enum class Color { Red, Black };

  /// Plain keyed node (reference BaseNode, rb_tree.hpp:41-74).
  template <KeyConcept Key> class BaseNode {
  protected:
    using key_type = std::remove_cv_t<Key>;
    using pointer = std::unique_ptr<BaseNode>;
    using reference_pointer = pointer &;
    using raw_pointer = BaseNode *;

  public:
    constexpr BaseNode() = default;
    BaseNode(BaseNode &&) noexcept = default;
    auto operator=(BaseNode &&) noexcept -> BaseNode & = default;
    explicit constexpr BaseNode(key_type value) : key{value} {}
    explicit constexpr BaseNode(Color color) : color_{color} {}
    constexpr BaseNode(key_type value, Color color) : key{value}, color_{color} {}
    virtual ~BaseNode() = default;

    void set_color(Color c) { color_ = c; }
    [[nodiscard]] auto is_black() const -> bool { return color_ == Color::Black; }
    [[nodiscard]] auto is_red() const -> bool { return color_ == Color::Red; }
    [[nodiscard]] auto leftr() const -> raw_pointer { return left.get(); }
    [[nodiscard]] auto rightr() const -> raw_pointer { return right.get(); }
    virtual void copy_key(const raw_pointer other) noexcept { key = other->key; }

    Key key{};
    Color color_{Color::Black};
    pointer left{nullptr};
    pointer right{nullptr};
    raw_pointer parent{nullptr};
  };

  class IntNode : public BaseNode<int> {
  public:
    using BaseNode::key_type;
    using BaseNode::pointer;
    using BaseNode::raw_pointer;
    using BaseNode::reference_pointer;
    constexpr IntNode() = default;
    using BaseNode::BaseNode;
    IntNode(IntNode &&) noexcept = default;
    IntNode &operator=(IntNode &&) noexcept = default;
    ~IntNode() override = default;
    auto operator<=>(IntNode const &o) const { return key <=> o.key; }
    friend bool operator==(IntNode const &a, IntNode const &b) { return a.key == b.key; }
  };

  namespace detail {
    /// Hook points a derived tree can use to maintain per-node augmentation (IntervalTree keeps `max`).
    struct NoAugment {
      template <typename N> static void on_descend(N *, N *) {}
      template <typename N> static void after_rotate(N * /*moved_down*/, N * /*moved_up*/) {}
    };
  }  // namespace detail

  template <NodeConcept NodeType, typename Augment = detail::NoAugment> class RbTree {
  public:
    using pointer = typename NodeType::pointer;
    using reference_pointer = typename NodeType::reference_pointer;
    using raw_pointer = typename NodeType::raw_pointer;

    constexpr RbTree() = default;
    RbTree(RbTree &&) noexcept = default;
    auto operator=(RbTree &&) noexcept -> RbTree & = default;
    RbTree(const RbTree &) = delete;
    auto operator=(const RbTree &) -> RbTree & = delete;
    virtual ~RbTree() { clear(); }

    /// Inserts every element of a range, in range order (reference rb_tree.hpp:111-117).
    template <std::ranges::input_range R>
      requires std::constructible_from<NodeType, std::ranges::range_value_t<R>>
    void insert_node(R &&range) {
      for (auto &&item : range) insert_node(std::forward<decltype(item)>(item));
    }
    /// Takes ownership of a node (reference :142-143).
    void insert_node(pointer node) { link_new(static_cast<raw_pointer>(node.release())); }
    /// Constructs the node in place (reference :145-149).
    template <typename... Args>
      requires std::constructible_from<NodeType, Args...>
    void insert_node(Args &&...args) {
      link_new(new NodeType(std::forward<Args>(args)...));
    }

    [[nodiscard]] auto empty() const -> bool { return root_ == nullptr; }
    [[nodiscard]] auto size() const -> std::size_t { return size(root()); }
    [[nodiscard]] auto size(raw_pointer n) const -> std::size_t {
      std::size_t count = 0;  // iterative: the reference recurses (rb_tree.hpp:177-180), the count is the same
      for (raw_pointer cur = n ? leftmost(n) : nullptr; cur != nullptr; cur = next_within(cur, n)) ++count;
      return count;
    }
    [[nodiscard]] auto root() const -> raw_pointer { return static_cast<raw_pointer>(root_.get()); }

    [[nodiscard]] auto minimum(raw_pointer n) const -> raw_pointer { return leftmost(n); }
    [[nodiscard]] auto maximum(raw_pointer n) const -> raw_pointer {
      while (n->right != nullptr) n = child(n->right);
      return n;
    }
    [[nodiscard]] auto successor(raw_pointer n) const -> raw_pointer {
      if (n->right != nullptr) return leftmost(child(n->right));
      raw_pointer up = up_of(n);
      while (up != nullptr && child(up->right) == n) {
        n = up;
        up = up_of(up);
      }
      return up;
    }
    [[nodiscard]] auto predecessor(raw_pointer n) const -> raw_pointer {
      if (n->left != nullptr) return maximum(child(n->left));
      raw_pointer up = up_of(n);
      while (up != nullptr && child(up->left) == n) {
        n = up;
        up = up_of(up);
      }
      return up;
    }
    [[nodiscard]] auto search(const typename NodeType::key_type &key) const -> raw_pointer {
      raw_pointer n = root();
      while (n != nullptr && !(key == n->key)) n = key < n->key ? child(n->left) : child(n->right);
      return n;
    }

    virtual void inorder_walk(raw_pointer n, int indent) const {
      if (n == nullptr) return;
      inorder_walk(child(n->left), indent);
      std::cout << std::string(static_cast<std::size_t>(indent < 0 ? 0 : indent), ' ') << n->key
                << (n->is_black() ? " B" : " R") << '\n';
      inorder_walk(child(n->right), indent);
    }

    /// Graphviz dump, pre-order (reference :402-411).
    void to_dot(std::string_view filename) const {
      std::ofstream out{std::string(filename)};
      out << "graph G {\nnode [fontname=\"Helvetica,Arial,sans-serif\"]\n";
      dot_subtree(out, root());
      out << "}\n";
    }

    /// CLRS delete (reference :506-557). Key-only: like the reference it does not repair augmentation.
    void delete_node(raw_pointer z) {
      if (z == nullptr) return;
      raw_pointer fix_parent = nullptr;  // parent of the (possibly null) node that took the removed black's place
      bool fix_is_left = false;
      bool removed_black = z->is_black();
      if (z->left == nullptr || z->right == nullptr) {
        raw_pointer p = up_of(z);
        fix_is_left = p != nullptr && child(p->left) == z;
        pointer sub = std::move(z->left != nullptr ? z->left : z->right);
        fix_parent = p;
        replace_slot(z, std::move(sub));  // destroys z
      } else {
        raw_pointer y = leftmost(child(z->right));
        removed_black = y->is_black();
        if (y == child(z->right)) {
          // y moves up into z's place and keeps its own right subtree
          fix_parent = y;
          fix_is_left = false;
          pointer ys = detach(y);  // owns y (with its right subtree); z->right is now empty
          ys->set_color(z->color_);
          ys->left = std::move(z->left);
          child(ys->left)->parent = ys.get();
          replace_slot(z, std::move(ys));
        } else {
          // deeper successor: its payload moves into z, then y itself is unlinked (reference copy_key :530)
          z->copy_key(y);
          fix_parent = up_of(y);
          fix_is_left = true;
          replace_slot(y, std::move(y->right));
        }
      }
      if (removed_black && fix_parent != nullptr) repair_after_delete(fix_parent, fix_is_left);
    }
    void delete_node(reference_pointer node) { delete_node(static_cast<raw_pointer>(node.get())); }

    void clear() noexcept {
      // unlink iteratively: default unique_ptr teardown would recurse once per level only, but keep it explicit
      root_.reset();
    }

  protected:
    static auto child(const pointer &p) -> raw_pointer { return static_cast<raw_pointer>(p.get()); }
    static auto up_of(raw_pointer n) -> raw_pointer { return static_cast<raw_pointer>(n->parent); }
    static auto is_red(raw_pointer n) -> bool { return n != nullptr && n->is_red(); }
    static auto leftmost(raw_pointer n) -> raw_pointer {
      while (n->left != nullptr) n = child(n->left);
      return n;
    }
    // in-order successor restricted to the subtree rooted at `top`
    auto next_within(raw_pointer n, raw_pointer top) const -> raw_pointer {
      if (n->right != nullptr) return leftmost(child(n->right));
      while (n != top) {
        raw_pointer p = up_of(n);
        if (child(p->left) == n) return p;
        n = p;
      }
      return nullptr;
    }

    /// The unique_ptr that owns `n` (root_ or a child slot of its parent).
    auto slot_of(raw_pointer n) -> pointer & {
      raw_pointer p = up_of(n);
      if (p == nullptr) return root_;
      return child(p->left) == n ? p->left : p->right;
    }
    /// Releases `n` from its owner and returns ownership to the caller.
    auto detach(raw_pointer n) -> pointer {
      pointer &s = slot_of(n);
      pointer out = std::move(s);
      out->parent = nullptr;
      return out;
    }
    /// Puts `with` where `n` hangs; `n` (and whatever it still owns) is destroyed.
    void replace_slot(raw_pointer n, pointer with) {
      raw_pointer p = up_of(n);
      pointer &s = slot_of(n);
      if (with != nullptr) with->parent = p;
      s = std::move(with);
    }

    // rotate_up(x): x's right (dir_left) / left child becomes x's parent. Returns the node that moved up.
    auto rotate(raw_pointer x, bool left_rotation) -> raw_pointer {
      raw_pointer p = up_of(x);
      pointer &xslot = slot_of(x);
      pointer xs = std::move(xslot);                                             // owns x
      pointer ys = std::move(left_rotation ? xs->right : xs->left);              // owns y
      assert(ys != nullptr);
      pointer &inner = left_rotation ? ys->left : ys->right;                     // y's inner subtree goes to x
      (left_rotation ? xs->right : xs->left) = std::move(inner);
      if (auto *moved = child(left_rotation ? xs->right : xs->left)) moved->parent = xs.get();
      raw_pointer xr = static_cast<raw_pointer>(xs.get());
      raw_pointer yr = static_cast<raw_pointer>(ys.get());
      xs->parent = yr;
      (left_rotation ? ys->left : ys->right) = std::move(xs);
      ys->parent = p;
      xslot = std::move(ys);
      Augment::after_rotate(xr, yr);
      return yr;
    }

    void link_new(raw_pointer z) {
      raw_pointer at = nullptr;
      for (raw_pointer cur = root(); cur != nullptr;) {
        at = cur;
        Augment::on_descend(cur, z);
        cur = z->key < cur->key ? child(cur->left) : child(cur->right);  // equal keys go right
      }
      z->parent = at;
      z->set_color(Color::Red);
      if (at == nullptr)
        root_.reset(z);
      else if (z->key < at->key)
        at->left.reset(z);
      else
        at->right.reset(z);
      repair_after_insert(z);
    }

    void repair_after_insert(raw_pointer z) {
      while (is_red(up_of(z))) {
        raw_pointer p = up_of(z), g = up_of(p);
        const bool p_is_left = child(g->left) == p;
        raw_pointer uncle = p_is_left ? child(g->right) : child(g->left);
        if (is_red(uncle)) {
          p->set_color(Color::Black);
          uncle->set_color(Color::Black);
          g->set_color(Color::Red);
          z = g;
          continue;
        }
        if ((p_is_left ? child(p->right) : child(p->left)) == z) {  // inner grandchild: straighten first
          rotate(p, p_is_left);
          z = p;
          p = up_of(z);
        }
        p->set_color(Color::Black);
        g->set_color(Color::Red);
        rotate(g, !p_is_left);
      }
      root()->set_color(Color::Black);
    }

    // `parent`'s left (is_left) or right subtree is one black short.
    void repair_after_delete(raw_pointer parent, bool is_left) {
      raw_pointer x = is_left ? child(parent->left) : child(parent->right);
      while (parent != nullptr && !is_red(x)) {
        raw_pointer w = is_left ? child(parent->right) : child(parent->left);
        if (w == nullptr) break;  // cannot happen in a valid red-black tree
        if (w->is_red()) {
          w->set_color(Color::Black);
          parent->set_color(Color::Red);
          rotate(parent, is_left);
          w = is_left ? child(parent->right) : child(parent->left);
          if (w == nullptr) break;
        }
        raw_pointer near = is_left ? child(w->left) : child(w->right);
        raw_pointer far = is_left ? child(w->right) : child(w->left);
        if (!is_red(near) && !is_red(far)) {
          w->set_color(Color::Red);
          x = parent;
          parent = up_of(x);
          if (parent != nullptr) is_left = child(parent->left) == x;
          continue;
        }
        if (!is_red(far)) {
          near->set_color(Color::Black);
          w->set_color(Color::Red);
          rotate(w, !is_left);
          w = is_left ? child(parent->right) : child(parent->left);
          far = is_left ? child(w->right) : child(w->left);
        }
        w->set_color(parent->color_);
        parent->set_color(Color::Black);
        far->set_color(Color::Black);
        rotate(parent, is_left);
        x = root();
        parent = nullptr;
      }
      if (x != nullptr) x->set_color(Color::Black);
    }

    virtual void dot_node(std::ofstream &out, raw_pointer n) const {
      out << n->key << " [label=\"" << n->key << "\", color=" << (n->is_black() ? "black" : "red")
          << ", style=bold];\n";
    }
    void dot_subtree(std::ofstream &out, raw_pointer n) const {
      if (n == nullptr) return;
      dot_node(out, n);
      dot_subtree(out, child(n->left));
      dot_subtree(out, child(n->right));
      if (n->left != nullptr) out << n->key << " -- " << child(n->left)->key << " [style=bold];\n";
      if (n->right != nullptr) out << n->key << " -- " << child(n->right)->key << " [style=bold];\n";
    }

    pointer root_{nullptr};
  };

}  // namespace binary::algorithm::tree

#endif  // BINARY_AMD_ALGORITHM_RB_TREE_HPP_
