"""pgrtk_amd -- MI355X-native SHIMMER indexing / query engine (host side mirror of pgr-tk's `pgrtk`).

All compute happens in libpgrhip.so (hand-written gfx950 HIP kernels behind the C ABI of
include/pgr_hip.h).  There is no CPU path: without the built library and a gfx950 device the
calls raise.
"""
from ._ffi import FRAG_REC, HITPAIR, MM128, Context, PackedSeqs, PgrError, Spec, default_context  # noqa: F401
from .engine import (Batch, Index, PackedBases, PinnedArrays, Pipe, Shmmrs, frag_recs_batch, make_spec, pack_ascii, records_checksum,  # noqa: F401
                     sequence_to_shmmrs, sequence_to_shmmrs_batch, sequence_to_shmmrs_batch_packed, time_shmmr_batch,
                     time_shmmr_batch_packed)
from .seqindexdb import (SeqIndexDB, get_shmmr_dots, get_shmmr_pairs_from_seq, pgr_lib_version, read_fastx,  # noqa: F401
                         sparse_aln, sparse_aln_groups)
from . import cli, mapgraph  # noqa: F401
from .helpers import (get_principle_bundle_bed_file_for_query, group_smps_by_principle_bundle_id, merge_regions, query_sdb, rc, rc_byte_seq, rc_u8_seq,  # noqa: F401
                      string_to_u8, u8_to_string)

__version__ = "0.4.0"
