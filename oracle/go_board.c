/*
 * go_board.c -- restatement of Gomoku::Board (TEST INFRASTRUCTURE, see gomoku_oracle.h).
 * Follows core/lib/src/Game.cpp:22-146 and core/py_ext/src/game_ext.hpp:87-104.
 */
#include "gomoku_oracle.h"
#include <string.h>

#define ST(b, player) ((b)->states[(player) + 1])
#define CNT(b, player) ((b)->counts[(player) + 1])

/* Game.cpp:22-30 */
static void set_state(go_board *b, int player, int pos) { ST(b, player)[pos] = 1; CNT(b, player) += 1; }
static void unset_state(go_board *b, int player, int pos) { ST(b, player)[pos] = 0; CNT(b, player) -= 1; }

/* Game.cpp:138-146 */
void go_board_reset(go_board *b) {
    for (int p = -1; p <= 1; ++p) {
        memset(ST(b, p), p == GO_NONE ? 1 : 0, GO_N);
        CNT(b, p) = (p == GO_NONE ? GO_N : 0);
    }
    b->nrec = 0;
    b->cur_player = GO_BLACK;
    b->winner = GO_NONE;
}

/* Game.cpp:80-82 */
int go_board_check_move(const go_board *b, int move) {
    return move >= 0 && move < GO_N && ST(b, GO_NONE)[move];
}

/* Game.cpp:37-47 */
int go_board_apply(go_board *b, int move, int check_victory) {
    if (b->cur_player != GO_NONE && go_board_check_move(b, move)) {
        set_state(b, b->cur_player, move);
        unset_state(b, GO_NONE, move);
        b->record[b->nrec++] = (int16_t)move;
        b->cur_player = (int8_t)-b->cur_player;
        if (check_victory) go_board_check_end(b);
    }
    return b->cur_player;
}

/* Game.cpp:49-62 */
int go_board_revert(go_board *b, int count) {
    if (b->cur_player == GO_NONE && count != 0) {
        b->cur_player = CNT(b, GO_BLACK) == CNT(b, GO_WHITE) ? GO_BLACK : GO_WHITE;
        b->winner = GO_NONE;
    }
    for (int i = 0; b->nrec > 0 && i < count; ++i) {
        int last = b->record[b->nrec - 1];
        unset_state(b, -b->cur_player, last);
        set_state(b, GO_NONE, last);
        b->nrec--;
        b->cur_player = (int8_t)-b->cur_player;
    }
    return b->cur_player;
}

/* Game.cpp:64-73 : the draw r replaces rnd(rnd_eng); the probe rule is the contract. */
int go_board_random_move(const go_board *b, unsigned r) {
    if (CNT(b, GO_NONE) == 0) return -1;       /* reference: throw overflow_error */
    int id = (int)(r % GO_N);
    while (!ST(b, GO_NONE)[id]) id = (id + 1) % GO_N;
    return id;
}

/* Game.cpp:84-86 */
static int check_boundary(int x, int y) { return x >= 0 && x < GO_W && y >= 0 && y < GO_H; }

/* Game.cpp:106-122 */
static int search_dir(const go_board *b, int cx, int cy, int last_player, int dx, int dy) {
    int renju = 1;
    for (int s = 0; s < 2; ++s) {
        int sgn = s == 0 ? 1 : -1;
        int x = cx, y = cy;
        for (int i = 1; i <= GO_RENJU; ++i) {
            x += sgn * dx; y += sgn * dy;
            if (check_boundary(x, y) && ST(b, last_player)[y * GO_W + x]) ++renju;
            else break;
        }
    }
    return renju >= GO_RENJU;
}

/* Game.cpp:88-136 */
int go_board_check_end(go_board *b) {
    if (b->cur_player == GO_NONE) return 1;
    if (b->nrec == 0) return 0;
    int last = b->record[b->nrec - 1];
    int cx = last % GO_W, cy = last / GO_W;
    int last_player = -b->cur_player;
    if (search_dir(b, cx, cy, last_player, 1, 0) || search_dir(b, cx, cy, last_player, 0, 1) ||
        search_dir(b, cx, cy, last_player, 1, -1) || search_dir(b, cx, cy, last_player, 1, 1)) {
        b->winner = (int8_t)last_player;
        b->cur_player = GO_NONE;
        return 1;
    } else if (CNT(b, GO_NONE) == 0) {
        b->winner = GO_NONE;
        b->cur_player = GO_NONE;
        return 1;
    }
    return 0;
}

/* game_ext.hpp:87-104 : [cur stones, opponent stones, empties, last move, move before last, black-to-move] */
void go_board_encoded_states(const go_board *b, uint8_t *out) {
    int planes[3] = { b->cur_player, -b->cur_player, GO_NONE };
    int index = 0;
    for (int k = 0; k < 3; ++k, ++index) memcpy(out + index * GO_N, ST(b, planes[k]), GO_N);
    for (int i = 0; i <= 1; ++index, ++i) {
        memset(out + index * GO_N, 0, GO_N);
        if (b->nrec > i) out[index * GO_N + b->record[b->nrec - 1 - i]] = 1;
    }
    memset(out + index * GO_N, b->cur_player == GO_BLACK, GO_N);
}

/* Test aid: replays n recorded games move by move through Board::applyMove with its victory check (Game.cpp:37-62, 88-136).
   For game g: legal[g] = 1 iff every move was accepted and none was played after the end; end_ply[g] = number of moves after which the
   board was over (-1: never); winner[g] = m_winner then.  Lets a full-size batch of records be checked in seconds. */
void go_board_replay_games(const uint8_t *moves, const int32_t *lens, int stride, int n, int8_t *legal, int32_t *end_ply, int8_t *winner) {
    for (int g = 0; g < n; ++g) {
        go_board b;
        go_board_reset(&b);
        int ok = 1, ended = -1;
        for (int i = 0; i < lens[g]; ++i) {
            if (b.cur_player == GO_NONE) { ok = 0; break; }                 /* a move after the end */
            int before = b.cur_player;
            int next = go_board_apply(&b, moves[(size_t)g * stride + i], 1);
            if (next == before) { ok = 0; break; }                          /* rejected: occupied or off the board (Game.cpp:38-39) */
            if (next == GO_NONE && ended < 0) ended = i + 1;
        }
        legal[g] = (int8_t)ok;
        end_ply[g] = ended;
        winner[g] = b.winner;
    }
}
