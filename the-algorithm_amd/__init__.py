"""MI355X-native SimClusters-ANN / representation-scorer hot path.

The directory name carries the reference's name (`the-algorithm_amd`), which is not a valid
Python identifier: import it through `tests/_pkg.py` / `__graft_entry__.load_package()`, which
register it as module `the_algorithm_amd`.
"""
from . import ann_codec, corpus, dense_ann, hnsw_ann, representation_scorer, sharding, simclusters_ann  # noqa: F401
from .simclusters_ann import (  # noqa: F401
    ApproximateCosineSimilarity,
    ClusterTweetIndex,
    LegacySimClustersANNCandidateSource,
    LegacySimClustersANNConfig,
    MicroBatcher,
    QueryBatch,
    ScoringAlgorithm,
    SimClustersANNConfig,
    Variant,
    load_library,
)
