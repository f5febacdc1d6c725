"""Growing 2-D lattice of the SOM: topology on a NetworkX graph, prototype vectors in ONE array.

Host-side counterpart of the reference's graph handling (dbgsom/BaseSom.py): the reference keeps
every prototype as a node attribute and rebuilds ``weights_`` with a Python list comprehension
and the hop matrix with Floyd-Warshall EVERY epoch (:397-401).  Here the prototypes live in a
contiguous (M, d) array in node order (what the device consumes), the hop matrix is a BFS
all-pairs computed only when the lattice changed, and the NetworkX node
attributes (``weight``, ``error``, ``epoch_created``) are written back on demand so ``som_`` looks
like the reference's.

Growth rules restated from the reference:
    distribute_errors   :563-586        add_new_neurons :588-614
    1 / 2 / 3 free positions :616-646 / :648-728 / :730-838     connect :840-861
"""
from __future__ import annotations

import networkx as nx
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import shortest_path


def _offsets():
    # order in which the reference probes the four lattice positions around a node
    return ((0, 1), (0, -1), (1, 0), (-1, 0))


class GrowingLattice:
    def __init__(self, init_vectors: np.ndarray):
        """2 x 2 start square with the given four prototype vectors (BaseSom.py:419-444)."""
        g = nx.Graph()
        corners = [(0, 0), (0, 1), (1, 0), (1, 1)]
        g.add_nodes_from(corners)
        g.add_edges_from([((0, 0), (0, 1)), ((0, 0), (1, 0)), ((1, 0), (1, 1)), ((0, 1), (1, 1))])
        self.graph = g
        self.W = np.array(init_vectors)  # keeps the sample dtype until the first update
        self.error = np.zeros(4)
        self.has_error = np.zeros(4, dtype=bool)  # the start nodes carry no "error" attribute yet
        self.epoch_created = np.zeros(4, dtype=np.int64)
        self._index = {n: i for i, n in enumerate(corners)}
        self._hops = None
        self._overwritten = []  # rows of already occupied positions a growth step rewrote

    # -- views ----------------------------------------------------------------------------------
    @property
    def nodes(self):
        return list(self.graph.nodes)

    def __len__(self):
        return self.graph.number_of_nodes()

    def index_of(self, node):
        return self._index[node]

    def weight(self, node):
        return self.W[self._index[node]]

    def hop_distances(self) -> np.ndarray:
        """All-pairs hop counts in node order, ``inf`` between components -- the values
        ``nx.floyd_warshall_numpy`` gives for this unweighted graph (BaseSom.py:367,401)."""
        if self._hops is None:
            idx = self._index
            m = len(idx)
            if self.graph.number_of_edges():
                u, v = zip(*[(idx[a], idx[b]) for a, b in self.graph.edges])
            else:
                u, v = (), ()
            adj = csr_matrix((np.ones(len(u)), (u, v)), shape=(m, m))
            self._hops = shortest_path(adj, method="D", directed=False, unweighted=True)
        return self._hops

    # -- per-epoch state ------------------------------------------------------------------------
    def set_weights(self, W_new: np.ndarray):
        self.W = W_new

    def set_errors(self, errors: np.ndarray):
        self.error = np.array(errors, dtype=np.float64)
        self.has_error[:] = True

    # -- growth ---------------------------------------------------------------------------------
    def distribute_errors(self, threshold: float):
        """Interior neurons above the threshold hand half of their error to their boundary
        neighbours, sequentially in node order (later nodes see earlier hand-overs)."""
        g, idx, err = self.graph, self._index, self.error
        for node, nbrs in g.adj.items():
            if len(nbrs) < 4:
                continue
            e = err[idx[node]]
            if not e > threshold:
                continue
            rim = [nb for nb in nbrs if len(g.adj[nb]) < 4]
            for nb in rim:
                err[idx[nb]] += 0.5 * e / len(rim)
            err[idx[node]] /= 2

    def will_grow(self, threshold: float) -> bool:
        """Whether `grow` would insert anything: its first candidate decides (it stops at the
        first one that is interior or below the threshold)."""
        i = int(np.argsort(-self.error)[0])
        return bool(self.error[i] > threshold and self.graph.degree(self.nodes[i]) < 4)

    def pop_overwritten(self):
        rows, self._overwritten = sorted(set(self._overwritten)), []
        return rows

    def grow(self, threshold: float, epoch: int) -> int:
        """Insert neurons next to boundary neurons whose error exceeds the threshold, largest
        error first; stops at the first candidate that is interior or below the threshold."""
        g = self.graph
        snapshot = self.error.copy()
        order = np.argsort(-snapshot)
        added = 0
        for i in order:
            node = self.nodes[i]
            deg = g.degree(node)
            if not (snapshot[i] > threshold and deg < 4):
                break
            if deg == 3:
                pos, w = self._one_free(node)
            elif deg == 2:
                pos, w = self._two_free(node)
            elif deg == 1:
                pos, w = self._three_free(node)
            else:
                continue
            self._insert(pos, w, epoch)
            added += 1
        return added

    def _one_free(self, node):
        x, y = node
        nbrs = self.graph.adj[node]
        for dx, dy in _offsets():
            cand = (x + dx, y + dy)
            if cand not in nbrs:
                opposite = (x - dx, y - dy)
                pos, w = cand, 2 * self.weight(node) - self.weight(opposite)
        return pos, w

    def _two_free(self, bo):
        n1, n2 = self.graph.adj[bo]
        e1, e2 = self.error[self._index[n1]], self.error[self._index[n2]]
        bx, by = bo
        away_from = n2 if e1 > e2 else n1  # grow opposite to the lower-error neighbour
        pos = (2 * bx - away_from[0], 2 * by - away_from[1])
        w = 2 * self.weight(bo) - self.weight(away_from)
        if n1[0] == n2[0] or n1[1] == n2[1]:  # the two neighbours face each other
            if n1[0] == n2[0]:
                pos, w = (bx + 1, by), 2 * self.weight(bo) - self.weight(n2)
            else:
                pos, w = (bx, by + 1), 2 * self.weight(bo) - self.weight(n1)
        return pos, w

    def _three_free(self, bo):
        bx, by = bo
        diagonal = {(bx + 1, by + 1), (bx + 1, by - 1), (bx - 1, by + 1), (bx - 1, by - 1)}
        n1 = list(self.graph.neighbors(bo))[0]
        side = list(diagonal.intersection(set(self.graph.neighbors(n1))))
        err = lambda n: self.error[self._index[n]]  # noqa: E731
        if len(side) == 0:
            return self._straight(n1, bo)
        if len(side) == 1:
            return self._bend(n1, bo, side[0])
        n2, n3 = side[0], side[1]
        if err(n1) > err(n2) and err(n1) > err(n3):
            return self._straight(n1, bo)
        return self._bend(n1, bo, n2 if err(n2) > err(n3) else n3)

    def _straight(self, neighbor, node):
        pos = (2 * node[0] - neighbor[0], 2 * node[1] - neighbor[1])
        return pos, 2 * self.weight(node) - self.weight(neighbor)

    def _bend(self, n1, bo, n2):
        if self.error[self._index[n1]] > self.error[self._index[n2]]:
            return self._straight(n1, bo)
        pos = (n2[0] + bo[0] - n1[0], n2[1] + bo[1] - n1[1])
        w = ((2 * self.weight(bo) - self.weight(n1)) + self.weight(n2)) / 2
        return pos, w

    def _insert(self, pos, w, epoch):
        g = self.graph
        if pos in self._index:  # the reference overwrites the attributes of an occupied position
            i = self._index[pos]
            self.W[i] = w
            self._overwritten.append(i)
            self.error[i] = 0.0
            self.epoch_created[i] = epoch
        else:
            g.add_node(pos)
            self._index[pos] = len(self._index)
            self.W = np.vstack([self.W, np.asarray(w)[None, :]])
            self.error = np.append(self.error, 0.0)
            self.has_error = np.append(self.has_error, True)
            self.epoch_created = np.append(self.epoch_created, epoch)
        x, y = pos
        fresh = []
        for nb in ((x, y + 1), (x, y - 1), (x - 1, y), (x + 1, y)):
            if nb in g.nodes:
                if not g.has_edge(pos, nb):
                    fresh.append(nb)
                g.add_edge(pos, nb)
        self._grow_hops(pos, fresh)

    def _grow_hops(self, pos, fresh):
        """Keep the all-pairs hop counts current across an insertion instead of recomputing them (a
        breadth-first search from every node, 1.3 ms at 250 neurons, per growth step): a NEW last node v
        with neighbours S has d(v, x) = 1 + min_s d(s, x), and a shortest path that did not exist before
        passes through v exactly once, d'(x, y) = min(d(x, y), d(x, v) + d(v, y)) -- the same integers.
        A fresh array every time (the backend keys its copy on the array's identity)."""
        H = self._hops
        if H is None:
            return                                  # nothing to keep: recomputed when next asked for
        m, i = H.shape[0], self._index[pos]
        if m == len(self._index):                   # an occupied position was rewritten
            if fresh:
                self._hops = None                   # (an edge nobody had drawn yet: recompute)
            return
        if i != m or m != len(self._index) - 1:     # (not a new LAST node: recompute)
            self._hops = None
            return
        rows = [self._index[nb] for nb in fresh]
        dv = 1.0 + H[rows].min(axis=0) if rows else np.full(m, np.inf)
        Hn = np.empty((m + 1, m + 1))
        np.minimum(H, dv[:, None] + dv[None, :], out=Hn[:m, :m])
        Hn[m, :m] = dv
        Hn[:m, m] = dv
        Hn[m, m] = 0.0
        self._hops = Hn

    # -- pruning / export -----------------------------------------------------------------------
    def remove(self, dead_nodes):
        """Drop nodes (dead neurons after the fit, BaseSom.py:223-235); order of the rest kept."""
        dead = set(dead_nodes)
        if not dead:
            self._hops = None  # the reference recomputes regardless
            return
        keep = [i for i, n in enumerate(self.nodes) if n not in dead]
        g = self.graph.copy()
        g.remove_nodes_from(dead)
        self.graph = g
        self.W = self.W[keep]
        self.error = self.error[keep]
        self.has_error = self.has_error[keep]
        self.epoch_created = self.epoch_created[keep]
        self._index = {n: i for i, n in enumerate(g.nodes)}
        self._hops = None

    def write_attributes(self, extra=None):
        """Mirror the arrays into node attributes so ``som_`` reads like the reference's graph."""
        g = self.graph
        for i, n in enumerate(g.nodes):
            attrs = g.nodes[n]
            attrs["weight"] = self.W[i]
            attrs["epoch_created"] = int(self.epoch_created[i])
            if self.has_error[i]:
                attrs["error"] = self.error[i]
            if extra:
                for key, arr in extra.items():
                    attrs[key] = arr[i]
        return g
