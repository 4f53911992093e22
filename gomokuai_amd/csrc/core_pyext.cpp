// core_pyext.cpp -- the `CorePyExt` Python module of the MI355X path.
//
// Mirrors, name for name, the pybind11 surface of the reference (core/py_ext/src/module.cpp:7-13,
// game_ext.hpp:13-106, mcts_ext.hpp:12-88, policy_ext.hpp:11-54) so that agents/mcts.py, agents/utils.py
// and network/data_helper.py run unchanged on top of it.  This file is the only code that touches Python
// objects; searches go through the C-ABI of libgomoku_hip.so (include/gomoku_hip.h) and run on the GPU.
// Board / Position / Player are plain host-side value types (the reference's are too): game bookkeeping,
// not the hot path.  There is no CPU search: without a GPU, MCTS.get_action / eval_state raise.
#include <pybind11/chrono.h>
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <chrono>
#include <cstring>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "../../include/gomoku_hip.h"

namespace py = pybind11;
using namespace py::literals;
using std::chrono::milliseconds;

namespace {

constexpr int kW = 15, kH = 15, kN = 225, kRenju = 5;

enum class Player : short { White = -1, None = 0, Black = 1 };
constexpr Player operator-(Player p) { return Player(-static_cast<short>(p)); }
constexpr float calc_score(Player player, Player winner) { return static_cast<float>(player) * static_cast<float>(winner); }
constexpr float calc_score(Player player, float value) { return static_cast<float>(player) * value; }

struct Position {
    short id;
    Position(int id_ = -1) : id(static_cast<short>(id_)) {}
    Position(int x, int y) : id(static_cast<short>(y * kW + x)) {}
    int x() const { return id % kW; }
    int y() const { return id / kW; }
    bool operator==(const Position& o) const { return id == o.id; }
};

std::string to_string(Position p) {
    std::ostringstream os;
    os << "(" << p.x() << ", " << p.y() << ")";
    return os.str();
}
std::string to_string(Player p) { return p == Player::Black ? "Player Black" : p == Player::White ? "Player White" : "No Player"; }

std::mt19937& rng() {                                   // reference: random_device-seeded engine (Game.cpp:11-12)
    static std::mt19937 eng((std::random_device())());
    return eng;
}
uint64_t g_search_seed = (static_cast<uint64_t>(std::random_device()()) << 32) | std::random_device()();
uint32_t g_next_game_id = 0;
// Default::AddNoise(m_root) runs at the start of every search with these defaults (MCTS.cpp:182, MonteCarlo.hpp:97); alpha 0 switches it off
float g_root_noise_alpha = 0.05f, g_root_noise_epsilon = 0.25f;

// ---- Board: same observable behaviour as Gomoku::Board (core/lib/src/Game.cpp:37-146) ----
class Board {
public:
    Board() { reset(); }

    Player apply_move(Position move, bool check_victory = true) {
        if (cur_ != Player::None && check_move(move)) {
            cell_[move.id] = cur_;
            counts_[idx(cur_)]++; counts_[idx(Player::None)]--;
            record_.push_back(move);
            cur_ = -cur_;
            if (check_victory) check_end();
        }
        return cur_;
    }

    Player revert_move(size_t count = 1) {
        if (cur_ == Player::None && count != 0) {
            cur_ = counts_[idx(Player::Black)] == counts_[idx(Player::White)] ? Player::Black : Player::White;
            winner_ = Player::None;
        }
        for (size_t i = 0; !record_.empty() && i < count; ++i) {
            const Position last = record_.back();
            counts_[idx(cell_[last.id])]--; counts_[idx(Player::None)]++;
            cell_[last.id] = Player::None;
            record_.pop_back();
            cur_ = -cur_;
        }
        return cur_;
    }

    Position random_move() const {
        if (counts_[idx(Player::None)] == 0) throw std::overflow_error("board is already full");
        int id = static_cast<int>(std::uniform_int_distribution<unsigned>(0, kN - 1)(rng()));
        while (cell_[id] != Player::None) id = (id + 1) % kN;
        return Position(id);
    }

    bool check_move(Position m) const { return m.id >= 0 && m.id < kN && cell_[m.id] == Player::None; }

    bool check_end() {
        if (cur_ == Player::None) return true;
        if (record_.empty()) return false;
        const int cx = record_.back().x(), cy = record_.back().y();
        const Player last = -cur_;
        auto run = [&](int dx, int dy) {
            int n = 1;
            for (int s : {1, -1})
                for (int i = 1, x = cx + s * dx, y = cy + s * dy; i <= kRenju; ++i, x += s * dx, y += s * dy) {
                    if (x >= 0 && x < kW && y >= 0 && y < kH && cell_[y * kW + x] == last) ++n; else break;
                }
            return n >= kRenju;
        };
        if (run(1, 0) || run(0, 1) || run(1, -1) || run(1, 1)) { winner_ = last; cur_ = Player::None; return true; }
        if (counts_[idx(Player::None)] == 0) { winner_ = Player::None; cur_ = Player::None; return true; }
        return false;
    }

    void reset() {
        cell_.fill(Player::None);
        counts_ = {0, kN, 0};
        record_.clear();
        cur_ = Player::Black;
        winner_ = Player::None;
    }

    // uint16[2][16] bit-planes for the C-ABI (black = plane 0)
    void planes(uint16_t out[32]) const {
        std::memset(out, 0, 64);
        for (int i = 0; i < kN; ++i)
            if (cell_[i] != Player::None) out[(cell_[i] == Player::Black ? 0 : 16) + i / kW] |= static_cast<uint16_t>(1u << (i % kW));
    }

    static int idx(Player p) { return static_cast<int>(p) + 1; }
    Player cur_ = Player::Black, winner_ = Player::None;
    std::array<Player, kN> cell_{};
    std::array<size_t, 3> counts_{};
    std::vector<Position> record_;
};

void throw_gmk(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("libgomoku_hip: ") + gmk_last_error());
}

// ---- Policy family (core/lib/include/MCTS.h:69-132, policies/*.h) ----
struct Node;
struct Policy {
    using SelectFunc = std::function<py::object(py::object)>;
    using ExpandFunc = std::function<size_t(py::object, Board&, py::array_t<float>)>;
    using EvalFunc = std::function<std::tuple<float, py::array_t<float>>(Board&)>;
    using UpdateFunc = std::function<void(py::object, Board&, double)>;
    Policy(SelectFunc s, ExpandFunc e, EvalFunc v, UpdateFunc u, double c) : select(s), expand(e), simulate(v), back_prop(u), c_puct(c) {}
    virtual ~Policy() = default;
    virtual const char* kind() const { return "Policy"; }
    bool has_python_stages() const { return select || expand || simulate || back_prop; }
    void prepare(Board& b) { init_acts = b.record_.size(); }
    void cleanup(Board& b) { b.revert_move(b.record_.size() - init_acts); }
    SelectFunc select; ExpandFunc expand; EvalFunc simulate; UpdateFunc back_prop;
    double c_puct;
    size_t init_acts = 0;
};
struct RandomPolicy : Policy {
    RandomPolicy(double c, size_t r) : Policy(nullptr, nullptr, nullptr, nullptr, c), c_rollouts(r) {}
    const char* kind() const override { return "RandomPolicy"; }
    size_t c_rollouts;
};
struct PoolRAVEPolicy : Policy {
    PoolRAVEPolicy(double c, double b) : Policy(nullptr, nullptr, nullptr, nullptr, c), c_bias(b) {}
    const char* kind() const override { return "PoolRAVEPolicy"; }
    double c_bias;
};
struct TraditionalPolicy : Policy {
    TraditionalPolicy(double c, double b, bool r) : Policy(nullptr, nullptr, nullptr, nullptr, c), c_bias(b), use_rave(r) {}
    const char* kind() const override { return "TraditionalPolicy"; }
    double c_bias; bool use_rave; size_t cached_acts = 0;
};

// ---- Node: value snapshot of a tree node (the reference hands out borrowed pointers that dangle after
// step_forward; a snapshot cannot dangle).  Children of the root carry the statistics of the last search. ----
struct Node {
    std::weak_ptr<Node> parent;                         // observer, like the reference's raw parent pointer
    Position position = Position(-1);
    Player player = Player::None;
    float state_value = 0.0f, action_prob = 0.0f;
    size_t node_visits = 0;
    std::vector<std::shared_ptr<Node>> children;
    bool is_leaf() const { return children.empty(); }
    bool is_full(const Board& b) const { return children.size() == b.counts_[Board::idx(Player::None)]; }
};

// ---- MCTS: one game searched on the GPU through gmk_mcts_* (core/lib/src/MCTS.cpp:60-198) ----
class MCTS {
public:
    MCTS(milliseconds duration, Position last_move, Player last_player, std::shared_ptr<Policy> policy)
        : policy_(policy ? policy : std::make_shared<RandomPolicy>(5.0, 5)), duration_(duration), by_iterations_(false) { init(last_move, last_player); }
    MCTS(size_t iterations, Position last_move, Player last_player, std::shared_ptr<Policy> policy)
        : policy_(policy ? policy : std::make_shared<RandomPolicy>(5.0, 5)), iterations_(iterations), duration_(0), by_iterations_(true) { init(last_move, last_player); }
    ~MCTS() { if (handle_) gmk_mcts_destroy(handle_); if (trad_handle_) gmk_trad_destroy(trad_handle_); if (az_handle_) gmk_az_destroy(az_handle_); }

    Position get_action(Board& board) {
        run_playouts(board);
        return step_forward_best();
    }

    std::tuple<float, py::array_t<float>> eval_state(Board& board) {
        run_playouts(board);
        py::array_t<float> pi(kN);
        throw_gmk(gmk_visits_to_pi(visits_.data(), static_cast<int>(board.record_.size()), pi.mutable_data()));
        return {root_->state_value, pi};
    }

    // MCTS::stepForward() (MCTS.cpp:129-134): the most visited child (first maximum) becomes the root
    Position step_forward_best() {
        std::shared_ptr<Node> best;
        for (auto& c : root_->children) if (!best || best->node_visits < c->node_visits) best = c;
        // TraditionalPolicy reorders children while it searches (RAVE::BackPropogate, MonteCarlo.hpp:179-181): the device
        // reports which of several equally visited children comes first in that order
        if (best_in_order_ >= 0)
            for (auto& c : root_->children) if (c->position.id == best_in_order_) best = c;
        best_in_order_ = -1;
        if (best) { best->parent.reset(); root_ = best; }
        return root_->position;
    }
    // MCTS::stepForward(move) (MCTS.cpp:136-147)
    void step_forward(Position move) {
        std::shared_ptr<Node> next;
        best_in_order_ = -1;
        for (auto& c : root_->children) if (c->position == move) { next = c; break; }
        if (!next) { next = std::make_shared<Node>(); next->position = move; next->player = -root_->player; next->action_prob = 1.0f; }
        next->parent.reset();
        root_ = next;
    }
    // MCTS::syncWithBoard (MCTS.cpp:119-125)
    void sync_with_board(Board& board) {
        size_t i = 0;
        while (i < board.record_.size() && !(board.record_[i] == root_->position)) ++i;
        i = (i == board.record_.size()) ? 0 : i + 1;
        for (; i < board.record_.size(); ++i) step_forward(board.record_[i]);
    }
    void reset() { init(Position(-1), Player::White); device_valid_ = false; trad_valid_ = false; az_valid_ = false; }

    std::shared_ptr<Policy> policy_;
    std::shared_ptr<Node> root_;
    size_t size_ = 1, iterations_ = 0;
    milliseconds duration_;

private:
    void init(Position last_move, Player last_player) {
        root_ = std::make_shared<Node>();
        root_->position = last_move; root_->player = last_player; root_->action_prob = 1.0f;
        size_ = 1;
        game_id_ = g_next_game_id++;
    }

    // Device-side bookkeeping of the RandomPolicy tree.  The tree lives on the GPU across calls, like the reference's
    // m_root: MCTS::syncWithBoard / stepForward(move) (MCTS.cpp:119-147) become gmk_mcts_step with the subtree kept, and
    // every search starts with Default::AddNoise on the root's children (MCTS.cpp:182; a no-op on a childless root).
    void device_fresh_root(Board& board) {
        uint16_t planes[32];
        board.planes(planes);
        const int16_t last = board.record_.empty() ? -1 : board.record_.back().id;
        throw_gmk(gmk_mcts_set_roots(handle_, planes, &last, game_id_));
        device_record_.assign(board.record_.begin(), board.record_.end());
    }

    // brings the device tree's root to the board's position: steps through the moves played since the last search when
    // the board continues that game, otherwise starts a fresh tree
    void device_sync(Board& board) {
        const size_t have = device_record_.size(), want = board.record_.size();
        bool continues = device_valid_ && have <= want;
        for (size_t i = 0; continues && i < have; ++i) continues = device_record_[i] == board.record_[i];
        if (!continues) { device_fresh_root(board); device_valid_ = true; return; }
        for (size_t i = have; i < want; ++i) {
            const int16_t mv = board.record_[i].id;
            throw_gmk(gmk_mcts_step_host(handle_, &mv, 1));
            device_record_.push_back(board.record_[i]);
        }
    }

    // MCTS::runPlayouts (MCTS.cpp:179-198)
    void run_playouts(Board& board) {
        const auto start = std::chrono::system_clock::now();
        auto* random = dynamic_cast<RandomPolicy*>(policy_.get());
        auto* trad = dynamic_cast<TraditionalPolicy*>(policy_.get());
        auto* rave = dynamic_cast<PoolRAVEPolicy*>(policy_.get());
        if (trad && !trad->use_rave && !policy_->has_python_stages()) { run_first_child_tree(board, trad->c_puct, trad, start); return; }
        if (rave && !policy_->has_python_stages()) { run_first_child_tree(board, rave->c_puct, nullptr, start); return; }
        if (!random && !trad && !rave && policy_->simulate && !policy_->select && !policy_->expand && !policy_->back_prop) { run_with_evaluator(board, start); return; }
        if (!random && !trad && !rave && (policy_->select || policy_->expand || policy_->back_prop)) { run_with_stages(board, start); return; }
        if (!random || policy_->has_python_stages())
            throw std::runtime_error(std::string("CorePyExt (MI355X): MCTS runs on the GPU with RandomPolicy, PoolRAVEPolicy, TraditionalPolicy(use_rave=False) and Policy(select=, expand=, eval_state=, back_prop=) in this build; ") +
                                     policy_->kind() + " is not one of them");
        throw_gmk(gmk_init(0));
        sync_with_board(board);
        policy_->prepare(board);
        const int chunk = by_iterations_ ? static_cast<int>(iterations_) : 256;
        // room for the kept subtree (at most everything the previous searches grew) plus this search
        const int capacity = by_iterations_ ? static_cast<int>(std::min<long long>(3ll * chunk * kN + 1, (1ll << 24) - 1)) : (1 << 22);
        if (!handle_ || capacity_ != capacity || c_rollouts_ != random->c_rollouts || c_puct_ != random->c_puct) {
            if (handle_) gmk_mcts_destroy(handle_);
            handle_ = nullptr;
            throw_gmk(gmk_mcts_create(1, capacity, random->c_puct, static_cast<int>(random->c_rollouts), g_search_seed, &handle_));
            capacity_ = capacity; c_rollouts_ = random->c_rollouts; c_puct_ = random->c_puct;
            device_valid_ = false;
        }
        device_sync(board);
        if (g_root_noise_alpha > 0.0f) throw_gmk(gmk_mcts_add_root_noise(handle_, g_root_noise_alpha, g_root_noise_epsilon, nullptr));
        uint32_t root_visits = 0, nodes = 1, nodes_before = 1;
        float q = 0.0f;
        int32_t status = 0;
        visits_.assign(kN, 0);
        throw_gmk(gmk_mcts_root_stats(handle_, visits_.data(), &q, &root_visits, &nodes_before, &status));     // the kept subtree
        if (by_iterations_) {
            throw_gmk(gmk_mcts_run(handle_, chunk, nullptr));
            throw_gmk(gmk_mcts_root_stats(handle_, visits_.data(), &q, &root_visits, &nodes, &status));
            duration_ = std::chrono::duration_cast<milliseconds>(std::chrono::system_clock::now() - start);
        } else {
            iterations_ = 0;
            for (auto end = start; end - start < duration_; end = std::chrono::system_clock::now()) {
                throw_gmk(gmk_mcts_run(handle_, chunk, nullptr));
                throw_gmk(gmk_mcts_root_stats(handle_, visits_.data(), &q, &root_visits, &nodes, &status));
                iterations_ += static_cast<size_t>(chunk);
                if (status & 2) break;                              // arena full
            }
        }
        // refresh the host-side view of the root and its children
        root_->state_value = q;
        root_->node_visits = root_visits;
        root_->children.clear();
        const size_t empties = board.counts_[Board::idx(Player::None)];
        if (nodes > 1)
            for (int i = 0; i < kN; ++i)
                if (board.cell_[i] == Player::None) {
                    auto c = std::make_shared<Node>();
                    c->parent = root_; c->position = Position(i); c->player = -root_->player;
                    c->action_prob = 1.0f / static_cast<float>(empties);
                    c->node_visits = visits_[i];
                    root_->children.push_back(c);
                }
        size_ += nodes - nodes_before;                              // m_size only ever grows (MCTS.cpp:189, 194)
        policy_->cleanup(board);
    }

    // the set-up of a search on the K7 tree (run_with_evaluator, run_with_stages): handle, MCTS::syncWithBoard, Default::AddNoise; returns the playout budget
    int az_prepare(Board& board) {
        throw_gmk(gmk_init(0));
        sync_with_board(board);
        policy_->prepare(board);
        const int budget = by_iterations_ ? static_cast<int>(iterations_) : (1 << 14);
        const int capacity = static_cast<int>(std::min<long long>(std::max<long long>(256, 3ll * budget * kN + 1), (1ll << 30)));       // the kept subtree plus this search
        if (!az_handle_ || az_capacity_ != capacity || az_c_puct_ != policy_->c_puct) {
            if (az_handle_) gmk_az_destroy(az_handle_);
            az_handle_ = nullptr;
            throw_gmk(gmk_az_create(1, capacity, policy_->c_puct, &az_handle_));
            az_capacity_ = capacity; az_c_puct_ = policy_->c_puct;
            az_valid_ = false;
        }
        // MCTS::syncWithBoard (MCTS.cpp:119-125): step through the moves played since the last search (subtree kept), or start over
        const size_t have = az_record_.size(), len = board.record_.size();
        bool continues = az_valid_ && have <= len;
        for (size_t i = 0; continues && i < have; ++i) continues = az_record_[i] == board.record_[i];
        if (!continues || have != len) az_playouts_done_ = 0;           // a new root: the playout numbers of Default::Simulate's streams start over (as GameHeader::playouts_done)
        if (continues) {
            for (size_t i = have; i < len; ++i) {
                const int16_t mv = board.record_[i].id;
                throw_gmk(gmk_az_step(az_handle_, &mv));
            }
        } else {
            uint16_t planes[32];
            board.planes(planes);
            const int16_t last[2] = {len >= 1 ? board.record_[len - 1].id : static_cast<short>(-1), len >= 2 ? board.record_[len - 2].id : static_cast<short>(-1)};
            throw_gmk(gmk_az_set_roots(az_handle_, planes, last));
        }
        az_record_.assign(board.record_.begin(), board.record_.end());
        az_valid_ = true;
        if (g_root_noise_alpha > 0.0f) throw_gmk(gmk_az_add_root_noise(az_handle_, g_root_noise_alpha, g_root_noise_epsilon, g_search_seed, game_id_));
        return budget;
    }

    // the root and its children as the search left them (value snapshots for MCTS.root); returns the tree's node count
    int32_t az_finish(Board& board) {
        std::vector<float> values(kN), priors(kN);
        uint32_t root_visits = 0;
        float q = 0.0f;
        int32_t nodes = 1, status = 0;
        visits_.assign(kN, 0);
        throw_gmk(gmk_az_root_stats(az_handle_, visits_.data(), values.data(), priors.data(), &root_visits, &q, &nodes, &status));
        if (status & 2)                                             // node arena full: playouts were dropped, the statistics are not those of the requested search
            throw std::overflow_error("CorePyExt (MI355X): the search tree outgrew its node arena (" + std::to_string(az_capacity_) + " nodes); playouts were dropped");
        root_->state_value = q;
        root_->node_visits = root_visits;
        root_->children.clear();
        for (int i = 0; i < kN; ++i)
            if (priors[i] != 0.0f && board.cell_[i] == Player::None) {
                auto c = std::make_shared<Node>();
                c->parent = root_; c->position = Position(i); c->player = -root_->player;
                c->action_prob = priors[i]; c->state_value = values[i]; c->node_visits = visits_[i];
                root_->children.push_back(c);
            }
        return nodes;
    }

    // MCTS(policy = Policy(eval_state = f, c_puct)) (agents/alphazero.py:5-9): the tree search runs on the device (K7, gmk_az_*, one
    // game), the evaluator is the Python callable, called once per playout with the leaf position like the reference's
    // policy->simulate(board) (MCTS.cpp:164-166).  The tree is kept from move to move and root noise is added like in the other searchers.
    void run_with_evaluator(Board& board, std::chrono::system_clock::time_point start) {
        const int budget = az_prepare(board);
        int32_t nodes_before = 1;
        throw_gmk(gmk_az_root_stats(az_handle_, nullptr, nullptr, nullptr, nullptr, nullptr, &nodes_before, nullptr));
        std::vector<int16_t> path(226);
        std::vector<float> probs(kN);
        size_t done = 0;
        auto playout = [&]() {
            int32_t depth = -1;
            throw_gmk(gmk_az_select_host(az_handle_, path.data(), &depth));
            if (depth >= 0) {                                       // a live leaf: ask the evaluator about that position
                Board leaf = board;
                for (int i = 0; i < depth; ++i) leaf.apply_move(Position(path[i]), false);        // Policy::applyMove: no victory check
                auto [value, arr] = policy_->simulate(leaf);
                auto flat = py::array_t<float, py::array::c_style | py::array::forcecast>(arr);
                if (flat.size() != kN) throw std::runtime_error("eval_state must return (value, probabilities of size 225)");
                std::memcpy(probs.data(), flat.data(), sizeof(float) * kN);
                throw_gmk(gmk_az_expand_host(az_handle_, &value, probs.data()));
            }
            ++done;
        };
        if (by_iterations_) {
            for (size_t i = 0; i < iterations_; ++i) playout();
            duration_ = std::chrono::duration_cast<milliseconds>(std::chrono::system_clock::now() - start);
        } else {
            for (auto end = start; end - start < duration_ && done < static_cast<size_t>(budget); end = std::chrono::system_clock::now()) playout();
            iterations_ = done;
        }
        const int32_t nodes = az_finish(board);
        size_ += static_cast<size_t>(nodes - nodes_before);
        policy_->cleanup(board);
    }

    // MCTS(policy = Policy(select=, expand=, eval_state=, back_prop=)) with Python callables in the TREE stages (SURVEY 8 a18;
    // core/py_ext/src/mcts_ext.hpp:43-61, MCTS::playout MCTS.cpp:149-177).  The callables run on the host, like every Python callable;
    // the tree stays on the device (K7's arena): per level the host reads the node and its children (gmk_az_read_*_host), hands `select`
    // value snapshots of them, tells the device which leaf the descent ended at, and the stages nobody replaced run where they always
    // run -- Default::Expand and Default::BackPropogate as az_expand_kernel's two halves, Default::Simulate's random rollout as a
    // one-wavefront kernel (its draws are those of gmk_mcts_*'s first rollout lane: Policy(select=Default::Select restated in Python) plays
    // MCTS(RandomPolicy(c_puct, 1))'s game).  A Python `expand` cannot add children in the reference either (Node.children is a copy,
    // create_node's result has no owner): it is called, its return value counts into MCTS.size, the node stays a leaf.
    void run_with_stages(Board& board, std::chrono::system_clock::time_point start) {
        const int budget = az_prepare(board);
        int32_t nodes_before = 1;
        throw_gmk(gmk_az_root_stats(az_handle_, nullptr, nullptr, nullptr, nullptr, nullptr, &nodes_before, nullptr));
        const uint32_t root_stones = static_cast<uint32_t>(board.record_.size());
        size_t done = 0, python_expanded = 0;
        struct Info { uint32_t visits = 0, parent = 0, first = 0; float value = 0, prior = 0; int32_t cell = -1, n = 0; };
        auto read_node = [&](uint32_t node) {
            Info i;
            throw_gmk(gmk_az_read_node_host(az_handle_, 0, node, &i.visits, &i.value, &i.prior, &i.cell, &i.parent, &i.first, &i.n));
            return i;
        };
        auto snapshot = [&](const Info& i, Player player, std::shared_ptr<Node> parent) {
            auto n = std::make_shared<Node>();
            n->parent = parent; n->position = Position(i.cell == 255 ? -1 : i.cell); n->player = player;
            n->state_value = i.value; n->action_prob = i.prior; n->node_visits = i.visits;
            return n;
        };
        auto playout = [&]() {
            // ---- select (MCTS.cpp:160-163) ----
            uint32_t node = 0;
            std::vector<int16_t> path;
            std::vector<uint32_t> path_nodes{0u};
            Board leaf = board;
            Player player = root_->player;
            for (;;) {
                const Info here = read_node(node);
                if (here.n == 0) break;                             // Node::isLeaf
                int16_t cells[kN]; uint32_t visits[kN]; float values[kN], priors[kN]; int32_t grand[kN];
                throw_gmk(gmk_az_read_children_host(az_handle_, 0, here.first, here.n, cells, visits, values, priors, grand));
                int pick = -1;
                if (policy_->select) {
                    auto obj = snapshot(here, player, nullptr);
                    for (int i = 0; i < here.n; ++i) {
                        Info ci; ci.visits = visits[i]; ci.value = values[i]; ci.prior = priors[i]; ci.cell = cells[i];
                        auto c = snapshot(ci, -player, obj);
                        c->children.resize(static_cast<size_t>(grand[i]));                 // (is_leaf / len(children) of a child: placeholders)
                        obj->children.push_back(c);
                    }
                    const py::object chosen = policy_->select(py::cast(obj));
                    const auto chosen_node = chosen.cast<std::shared_ptr<Node>>();
                    for (int i = 0; i < here.n && pick < 0; ++i) if (obj->children[static_cast<size_t>(i)] == chosen_node) pick = i;
                    for (int i = 0; i < here.n && pick < 0 && chosen_node; ++i) if (chosen_node->position.id == cells[i]) pick = i;
                    if (pick < 0) throw std::runtime_error("Policy.select must return one of the node's children");
                } else {                                            // Default::Select (MonteCarlo.hpp:57-68), as az_select_kernel scores
                    const double sqrt_n = std::sqrt(static_cast<double>(here.visits));
                    double best = -1.0;
                    pick = 0;
                    for (int i = 0; i < here.n; ++i) {
                        const double score = static_cast<double>(values[i]) + policy_->c_puct * static_cast<double>(priors[i]) * sqrt_n / static_cast<double>(visits[i] + 1u);
                        if (score > best) { best = score; pick = i; }
                    }
                }
                node = here.first + static_cast<uint32_t>(pick);
                path.push_back(cells[pick]);
                path_nodes.push_back(node);
                leaf.apply_move(Position(cells[pick]), false);      // Policy::applyMove: no victory check
                player = -player;
            }
            throw_gmk(gmk_az_set_leaf_host(az_handle_, 0, node, path.data(), static_cast<int>(path.size())));
            // ---- the leaf: finished game, or simulate + expand (MCTS.cpp:164-172) ----
            float state_value = 0.0f;                               // for the player to move at the leaf; the node's value is its negative
            double node_value;
            bool expand_on_device = false;
            std::vector<float> probs(kN, 0.0f);
            std::shared_ptr<Node> leaf_obj;
            if (leaf.check_end()) {
                node_value = calc_score(player, leaf.winner_);
            } else {
                if (policy_->simulate) {
                    auto [value, arr] = policy_->simulate(leaf);
                    auto flat = py::array_t<float, py::array::c_style | py::array::forcecast>(arr);
                    if (flat.size() != kN) throw std::runtime_error("eval_state must return (value, probabilities of size 225)");
                    std::memcpy(probs.data(), flat.data(), sizeof(float) * kN);
                    state_value = value;
                } else {                                            // Default::Simulate: one random rollout, uniform probabilities
                    int32_t winner = 0;
                    throw_gmk(gmk_az_rollout_host(az_handle_, 0, g_search_seed, game_id_, az_playouts_done_, root_stones << 8, &winner));
                    state_value = calc_score(leaf.cur_, static_cast<Player>(winner));
                    const float uniform = 1.0f / static_cast<float>(leaf.counts_[Board::idx(Player::None)]);
                    for (int i = 0; i < kN; ++i) probs[static_cast<size_t>(i)] = leaf.cell_[static_cast<size_t>(i)] == Player::None ? uniform : 0.0f;
                }
                if (policy_->expand) {
                    leaf_obj = snapshot(read_node(node), player, nullptr);
                    py::array_t<float> arr(kN);
                    std::memcpy(arr.mutable_data(), probs.data(), sizeof(float) * kN);
                    python_expanded += policy_->expand(py::cast(leaf_obj), leaf, arr);
                } else {
                    expand_on_device = true;
                }
                node_value = -static_cast<double>(state_value);
            }
            // ---- back-propagate (MCTS.cpp:173) ----
            if (policy_->back_prop) {
                throw_gmk(gmk_az_expand_stages_host(az_handle_, nullptr, probs.data(), expand_on_device ? 1 : 0, 0));
                std::vector<std::shared_ptr<Node>> chain;
                Player p = root_->player;
                for (size_t i = 0; i < path_nodes.size(); ++i, p = -p)
                    chain.push_back(snapshot(read_node(path_nodes[i]), p, i ? chain[i - 1] : nullptr));
                policy_->back_prop(py::cast(chain.back()), leaf, node_value);
                std::vector<uint32_t> v(chain.size());
                std::vector<float> q(chain.size());
                for (size_t i = 0; i < chain.size(); ++i) { v[i] = static_cast<uint32_t>(chain[i]->node_visits); q[i] = chain[i]->state_value; }
                throw_gmk(gmk_az_write_stats_host(az_handle_, 0, path_nodes.data(), v.data(), q.data(), static_cast<int>(chain.size())));
            } else {
                const float minus = static_cast<float>(-node_value);   // az_expand_kernel backs up the negative of what it is given
                throw_gmk(gmk_az_expand_stages_host(az_handle_, &minus, probs.data(), expand_on_device ? 1 : 0, 1));
            }
            ++az_playouts_done_;
            ++done;
        };
        if (by_iterations_) {
            for (size_t i = 0; i < iterations_; ++i) playout();
            duration_ = std::chrono::duration_cast<milliseconds>(std::chrono::system_clock::now() - start);
        } else {
            for (auto end = start; end - start < duration_ && done < static_cast<size_t>(budget); end = std::chrono::system_clock::now()) playout();
            iterations_ = done;
        }
        const int32_t nodes = az_finish(board);
        size_ += static_cast<size_t>(nodes - nodes_before) + python_expanded;
        policy_->cleanup(board);
    }

    // MCTS(policy = TraditionalPolicy) on the device (K6, gmk_trad_*): one game; the handle owns the policy's evaluator,
    // which persists across searches like TraditionalPolicy::m_evaluator (Traditional.h:27-31).  trad == nullptr:
    // MCTS(policy = PoolRAVEPolicy) on the same tree (K8, gmk_trad_run_poolrave; PoolRAVE.h:7-52)
    void run_first_child_tree(Board& board, double c_puct, TraditionalPolicy* trad, std::chrono::system_clock::time_point start) {
        auto run_chunk = [&](int n) {
            throw_gmk(trad ? gmk_trad_run(trad_handle_, n, c_puct, nullptr) : gmk_trad_run_poolrave(trad_handle_, n, c_puct, g_search_seed, game_id_, nullptr));
        };
        throw_gmk(gmk_init(0));
        sync_with_board(board);
        policy_->prepare(board);
        const int chunk = by_iterations_ ? static_cast<int>(iterations_) : 256;
        // room for the kept subtree plus this search
        const long long want = by_iterations_ ? 3ll * chunk * 226 + 1 : (1ll << 22);
        const int capacity = static_cast<int>(std::min<long long>(std::max<long long>(want, 256), (1ll << 24) - 1));
        if (!trad_handle_ || trad_capacity_ != capacity) {
            if (trad_handle_) gmk_trad_destroy(trad_handle_);
            trad_handle_ = nullptr;
            throw_gmk(gmk_trad_create(1, capacity, &trad_handle_));
            trad_capacity_ = capacity;
            trad_valid_ = false;
        }
        // MCTS::syncWithBoard (MCTS.cpp:119-125): step through the moves played since the last search, subtree kept
        const size_t have = trad_record_.size(), now = board.record_.size();
        bool continues = trad_valid_ && have <= now;
        for (size_t i = 0; continues && i < have; ++i) continues = trad_record_[i] == board.record_[i];
        if (continues) {
            for (size_t i = have; i < now; ++i) {
                const int16_t mv = board.record_[i].id;
                throw_gmk(gmk_trad_step(trad_handle_, &mv));
            }
        } else {
            uint8_t moves[kN] = {};
            const int32_t len = static_cast<int32_t>(now);
            for (int i = 0; i < len; ++i) moves[i] = static_cast<uint8_t>(board.record_[i].id);
            throw_gmk(gmk_trad_set_positions(trad_handle_, moves, &len));
        }
        trad_record_.assign(board.record_.begin(), board.record_.end());
        trad_valid_ = true;
        if (g_root_noise_alpha > 0.0f) throw_gmk(gmk_trad_add_root_noise(trad_handle_, g_root_noise_alpha, g_root_noise_epsilon, g_search_seed, game_id_));
        std::vector<float> values(kN), priors(kN);
        uint32_t root_visits = 0;
        float q = 0.0f;
        int32_t best = -1, nodes = 1, status = 0;
        visits_.assign(kN, 0);
        auto read = [&]() { throw_gmk(gmk_trad_root_stats(trad_handle_, visits_.data(), values.data(), priors.data(), &best, &root_visits, &q, &nodes, &status, nullptr)); };
        if (by_iterations_) {
            run_chunk(chunk);
            read();
            duration_ = std::chrono::duration_cast<milliseconds>(std::chrono::system_clock::now() - start);
        } else {
            iterations_ = 0;
            for (auto end = start; end - start < duration_; end = std::chrono::system_clock::now()) {
                run_chunk(chunk);
                read();
                iterations_ += static_cast<size_t>(chunk);
                if (status & 1) break;                              // node arena full
            }
        }
        if (status & 6) throw std::runtime_error("CorePyExt (MI355X): the device evaluator reported an inconsistent state");
        root_->state_value = q;
        root_->node_visits = root_visits;
        root_->children.clear();
        for (int i = 0; i < kN; ++i)
            if (priors[i] != 0.0f) {                                // Default::Expand: one child per cell with a non-zero prior
                auto c = std::make_shared<Node>();
                c->parent = root_; c->position = Position(i); c->player = -root_->player;
                c->action_prob = priors[i]; c->state_value = values[i]; c->node_visits = visits_[i];
                root_->children.push_back(c);
            }
        best_in_order_ = best;
        size_ = static_cast<size_t>(nodes);
        if (trad) trad->cached_acts = policy_->init_acts;
        policy_->cleanup(board);
    }

    gmk_mcts* handle_ = nullptr;
    std::vector<Position> device_record_;                           // the moves that lead to the device tree's root
    bool device_valid_ = false;
    gmk_trad* trad_handle_ = nullptr;
    gmk_az* az_handle_ = nullptr;
    std::vector<Position> az_record_;                               // the moves that lead to the device tree's root (Policy(eval_state=...))
    bool az_valid_ = false;
    uint32_t az_playouts_done_ = 0;                                 // playouts on the current root (the counter of Default::Simulate's draws)
    int az_capacity_ = 0;
    double az_c_puct_ = 0;
    int trad_capacity_ = 0, best_in_order_ = -1;
    std::vector<Position> trad_record_;                             // the moves that lead to the device tree's root (TraditionalPolicy)
    bool trad_valid_ = false;
    int capacity_ = 0;
    size_t c_rollouts_ = 0;
    double c_puct_ = 0;
    bool by_iterations_;
    uint32_t game_id_ = 0;
    std::vector<uint32_t> visits_ = std::vector<uint32_t>(kN, 0);
};

}  // namespace

PYBIND11_MODULE(CorePyExt, mod) {
    mod.doc() = "Gomoku AI core module (MI355X build: searches run on the GPU through libgomoku_hip)";

    mod.add_object("GameConfig", py::dict("width"_a = kW, "height"_a = kH, "board_size"_a = kN, "max_renju"_a = kRenju));
    // extension: make searches reproducible (the reference seeds from random_device and has no such hook)
    mod.def("set_root_noise", [](float alpha, float epsilon) { g_root_noise_alpha = alpha; g_root_noise_epsilon = epsilon; }, "alpha"_a = 0.05f, "epsilon"_a = 0.25f,
            "MI355X build only: the Dirichlet noise Default::AddNoise mixes into the root priors before every search (alpha 0 = none)");
    mod.def("set_seed", [](uint64_t seed) { g_search_seed = seed; g_next_game_id = 0; rng().seed(static_cast<uint32_t>(seed)); }, "seed"_a);

    py::enum_<Player>(mod, "Player", "Gomoku player types")
        .value("white", Player::White).value("none", Player::None).value("black", Player::Black)
        .def("__float__", [](Player p) { return static_cast<double>(p); })
        .def("__neg__", [](Player p) { return -p; })
        .def_static("calc_score", [](Player p, Player w) { return static_cast<double>(calc_score(p, w)); })
        .def_static("calc_score", [](Player p, double v) { return static_cast<double>(calc_score(p, static_cast<float>(v))); });

    py::class_<Position>(mod, "Position", "Gomoku board positions")
        .def(py::init<int>())
        .def(py::init<int, int>())
        .def_readwrite("id", &Position::id)
        .def_property("x", &Position::x, [](Position& p, int x) { p.id = static_cast<short>(p.y() * kW + x); })
        .def_property("y", &Position::y, [](Position& p, int y) { p.id = static_cast<short>(p.id + (y - p.y()) * kW); })
        .def("__int__", [](const Position& p) { return static_cast<int>(p.id); })
        .def("__index__", [](const Position& p) { return static_cast<int>(p.id); })
        .def("__hash__", [](const Position& p) { return static_cast<size_t>(p.id); })
        .def("__eq__", [](const Position& a, const Position& b) { return a.id == b.id; })
        .def("__len__", [](const Position&) { return 2; })
        .def("__repr__", [](const Position& p) { return "Position" + to_string(p); })
        .def("__str__", [](const Position& p) { return to_string(p); })
        .def("__iter__", [](const Position& p) {
            py::list items;
            items.append(py::make_tuple("x", p.x()));
            items.append(py::make_tuple("y", p.y()));
            return py::iter(items);
        });
    py::implicitly_convertible<int, Position>();

    py::class_<Board>(mod, "Board", "Gomoku game board")
        .def(py::init<>())
        .def("apply_move", &Board::apply_move, "move"_a, "checkVictory"_a = true)
        .def("revert_move", &Board::revert_move, "count"_a = 1)
        .def("random_move", &Board::random_move)
        .def("check_move", &Board::check_move)
        .def("check_end", &Board::check_end)
        .def("reset", &Board::reset)
        .def_property_readonly("move_record", [](const Board& b) { return b.record_; })
        .def_property_readonly("last_move", [](const Board& b) { return b.record_.empty() ? Position(-1) : b.record_.back(); })
        .def_property_readonly("move_counts", [](const Board& b) {
            py::dict d;
            for (Player p : {Player::Black, Player::None, Player::White}) d[py::cast(p)] = b.counts_[Board::idx(p)];
            return d;
        })
        .def_property_readonly("move_states", [](const Board& b) {
            py::dict d;
            for (Player p : {Player::Black, Player::None, Player::White}) {
                py::array_t<uint8_t> a({kH, kW});
                for (int i = 0; i < kN; ++i) a.mutable_data()[i] = b.cell_[i] == p;
                d[py::cast(p)] = a;
            }
            return d;
        })
        .def_property_readonly("status", [](const Board& b) {
            return py::dict("is_end"_a = (b.cur_ == Player::None), "cur_player"_a = b.cur_, "winner"_a = b.winner_);
        })
        .def("encoded_states", [](const Board& b) {                       // game_ext.hpp:87-104
            py::array_t<uint8_t> s({6, kH, kW});
            uint8_t* d = s.mutable_data();
            const Player planes[3] = {b.cur_, -b.cur_, Player::None};
            for (int k = 0; k < 3; ++k) for (int i = 0; i < kN; ++i) d[k * kN + i] = b.cell_[i] == planes[k];
            for (int k = 0; k < 2; ++k) {
                std::memset(d + (3 + k) * kN, 0, kN);
                if (b.record_.size() > static_cast<size_t>(k)) d[(3 + k) * kN + (b.record_.rbegin() + k)->id] = 1;
            }
            std::memset(d + 5 * kN, b.cur_ == Player::Black, kN);
            return s;
        }, "Feature planes: [X_t, Y_t, Z_t, y_t-1, x_t-2, C<is_black>]")
        .def("__repr__", [](const Board& b) { return "Board(cur_player: " + to_string(b.cur_) + ")"; })
        .def("__str__", [](const Board& b) {
            std::ostringstream os;
            os << std::hex << "  ";
            for (int x = 0; x < kW; ++x) os << x << " ";
            os << "\n";
            for (int y = 0; y < kH; ++y) {
                os << y << " ";
                for (int x = 0; x < kW; ++x) os << (b.cell_[y * kW + x] == Player::Black ? "x " : b.cell_[y * kW + x] == Player::White ? "o " : "_ ");
                os << "\n";
            }
            return os.str();
        });

    py::class_<Node, std::shared_ptr<Node>>(mod, "Node", "MCTS Tree Node")
        .def(py::init([](std::shared_ptr<Node> parent, Position position, Player player, float value, float prob) {
                 auto n = std::make_shared<Node>();
                 n->parent = parent; n->position = position; n->player = player; n->state_value = value; n->action_prob = prob;
                 return n;
             }), "parent"_a = nullptr, "position"_a = Position(-1), "player"_a = Player::None, "state_value"_a = 0.0, "action_prob"_a = 0.0)
        .def_property_readonly("parent", [](const Node& n) { return n.parent.lock(); })
        .def_readonly("position", &Node::position)
        .def_readonly("player", &Node::player)
        .def_readwrite("state_value", &Node::state_value)
        .def_readwrite("action_prob", &Node::action_prob)
        .def_readwrite("node_visits", &Node::node_visits)
        .def_property_readonly("children", [](const Node& n) { return n.children; })
        .def("is_leaf", &Node::is_leaf)
        .def("is_full", &Node::is_full)
        .def("__repr__", [](const Node& n) {
            std::ostringstream os;
            os << "Node(pose: " << to_string(n.position) << ", player: " << to_string(n.player) << ", value: " << n.state_value
               << ", prob: " << n.action_prob << ", visits: " << n.node_visits << ", childs: " << n.children.size() << ")";
            return os.str();
        });

    py::class_<Policy, std::shared_ptr<Policy>>(mod, "Policy", "MCTS Tree Policy")
        .def(py::init<Policy::SelectFunc, Policy::ExpandFunc, Policy::EvalFunc, Policy::UpdateFunc, double>(),
             "select"_a = nullptr, "expand"_a = nullptr, "eval_state"_a = nullptr, "back_prop"_a = nullptr, "c_puct"_a = 5.0)
        .def("prepare", &Policy::prepare)
        .def("clean_up", &Policy::cleanup)
        .def("apply_move", [](Policy&, Board& b, Position m) { return b.apply_move(m, false); })
        .def("revert_move", [](Policy&, Board& b, size_t count) { return b.revert_move(count); }, "board"_a, "count"_a = 1)
        .def("check_game_end", [](Policy&, Board& b) { return b.check_end(); })
        .def("create_node", [](Policy&, std::shared_ptr<Node> parent, Position pose, Player player, float value, float prob) {
            auto n = std::make_shared<Node>();
            n->parent = parent; n->position = pose; n->player = player; n->state_value = value; n->action_prob = prob;
            return n;
        })
        .def_readonly("select", &Policy::select)
        .def_readonly("expand", &Policy::expand)
        .def_readonly("eval_state", &Policy::simulate)
        .def_readonly("back_prop", &Policy::back_prop)
        .def_readonly("c_puct", &Policy::c_puct)
        .def("__repr__", [](const Policy& p) {
            std::ostringstream os;
            os << "Policy(c_puct: " << p.c_puct << ", init_acts: " << p.init_acts << ")";
            return os.str();
        });

    py::class_<RandomPolicy, Policy, std::shared_ptr<RandomPolicy>>(mod, "RandomPolicy", "Random policy with averaged mutliple rollouts")
        .def(py::init<double, size_t>(), "c_puct"_a = 5.0, "c_rollouts"_a = 5)
        .def_readonly("c_rollouts", &RandomPolicy::c_rollouts)
        .def("__repr__", [](const RandomPolicy& p) {
            std::ostringstream os;
            os << "RandomPolicy(c_puct: " << p.c_puct << ", c_rollouts: " << p.c_rollouts << ", init_acts: " << p.init_acts << ")";
            return os.str();
        });
    py::class_<PoolRAVEPolicy, Policy, std::shared_ptr<PoolRAVEPolicy>>(mod, "PoolRAVEPolicy", "PoolRAVE policy with MC-RAVE algorithm")
        .def(py::init<double, double>(), "c_puct"_a = 2, "c_bias"_a = 0)
        .def("__repr__", [](const PoolRAVEPolicy& p) {
            std::ostringstream os;
            os << "PoolRAVEPolicy(c_puct: " << p.c_puct << ", c_bias: " << p.c_bias << ", init_acts: " << p.init_acts << ")";
            return os.str();
        });
    py::class_<TraditionalPolicy, Policy, std::shared_ptr<TraditionalPolicy>>(mod, "TraditionalPolicy", "Traditional policy with MC + Pattern Matching algorithm")
        .def(py::init<double, double, bool>(), "c_puct"_a = 5.0, "c_bias"_a = 0, "use_rave"_a = false)
        .def("__repr__", [](const TraditionalPolicy& p) {
            std::ostringstream os;
            os << "TraditionalPolicy(c_puct: " << p.c_puct << ", init_acts: " << p.init_acts << ", cached_acts: " << p.cached_acts << ")";
            return os.str();
        });

    py::class_<MCTS>(mod, "MCTS", "Monte Carlo Tree Search")
        .def(py::init<milliseconds, Position, Player, std::shared_ptr<Policy>>(),
             "c_duration"_a = milliseconds(960), "last_move"_a = Position(-1), "last_player"_a = Player::White, py::arg_v("policy", nullptr, "Default Policy"))
        .def(py::init<size_t, Position, Player, std::shared_ptr<Policy>>(),
             "c_iterations"_a, "last_move"_a = Position(-1), "last_player"_a = Player::White, py::arg_v("policy", nullptr, "Default Policy"))
        .def_readonly("size", &MCTS::size_)
        .def_readonly("iterations", &MCTS::iterations_)
        .def_readonly("duration", &MCTS::duration_)
        .def_property_readonly("root", [](const MCTS& m) { return m.root_; })
        .def_property_readonly("policy", [](const MCTS& m) { return m.policy_; })
        .def("get_action", &MCTS::get_action)
        .def("eval_state", &MCTS::eval_state)
        .def("step_forward", [](MCTS& m) { m.step_forward_best(); })
        .def("step_forward", [](MCTS& m, Position p) { m.step_forward(p); }, "next_move"_a)
        .def("sync_with_board", &MCTS::sync_with_board)
        .def("reset", &MCTS::reset)
        .def("__repr__", [](const MCTS& m) {
            std::ostringstream os;
            os << "MCTS(root_player: " << to_string(m.root_->player) << ", nodes: " << m.size_ << ")";
            return os.str();
        });
}
