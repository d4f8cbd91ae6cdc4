"""MI355X-native BP-gauged gate application (drop-in for TensorNetworkQuantumSimulator.jl's apply_gates /
BeliefPropagationCache path).  See DESIGN.md and INTEGRATION.md at the repository root."""
from ._lib import LIB_PATH, EXPORTS, TnqsError, TnqsArgumentError, TnqsDomainError
from .graphs import (NamedGraph, named_grid, named_hexagonal_lattice_graph, heavy_hexagonal_lattice, named_comb_tree,
                     build_graph_from_gates, build_graph_from_circuit, edge_color, forest_cover_edge_sequence, steiner_region)
from .gates import (GATES, ALIASES, BUILTIN_GATES, gate_matrix, register_gate, register_alias, unregister_gate, levenshtein)
from .core import (TensorNetworkState, tensornetworkstate, random_tensornetworkstate, BeliefPropagationCache, network,
                   scalartype, maxvirtualdim, default_bp_update_kwargs, default_tolerance, update, apply_gates,
                   apply_circuit, truncate, expect, expect_all, rdm, vertex_scalars, edge_scalars, freenergy, partitionfunction,
                   rescale, rescale_messages, rescale_vertices, normalize, symmetric_gauge, symmetrize_and_normalize, profile_enable, profile_get, profile_reset,
                   PROF_CLASSES)
from . import dist
from .dist import partition_vertices, shard
