"""Per-request bookkeeping of block scores and of the batch-index -> (request, sequence)
map, with the interface of the reference's
vllm/attention/block_l2norm_manager.py:8-106 (same method names and return values).

State is kept per request id; a request that disappears from the batch is dropped and
starts from scratch if it comes back (reference :48-53).  Scores live on the host here;
the device-resident score table used by the decode kernel is managed by
``backend.DeviceScoreTable`` and addressed through the same flat map.
"""
from typing import Dict, List, Optional, Tuple


class BlockL2NormManager:

    def __init__(self) -> None:
        self.reqid_mapping_blocktables: Optional[Dict[str, Dict[int, List[int]]]] = None
        self.reqid_mapping_seqs: Optional[Dict[str, List[int]]] = None
        self.flat_sequence_map: List[Tuple[str, int]] = []
        self.reqid_mapping_block_l2norms: Optional[Dict[str, Dict[int, List[float]]]] = None

    # -- batch composition ---------------------------------------------------------------
    def update_reqs(self, reqid_mapping_blocktables: Dict[str, Dict[int, List[int]]]) -> bool:
        """Adopt this step's request -> {seq_id: block table} mapping; returns True when the
        set of requests changed (the flat map was rebuilt)."""
        first = self.reqid_mapping_block_l2norms is None
        if first:
            self.reqid_mapping_seqs = {}
            self.reqid_mapping_block_l2norms = {}
            changed = True
        else:
            changed = set(reqid_mapping_blocktables) != set(self.reqid_mapping_blocktables)
        self.reqid_mapping_blocktables = reqid_mapping_blocktables
        norms, seqs = self.reqid_mapping_block_l2norms, self.reqid_mapping_seqs
        for req_id, tables in reqid_mapping_blocktables.items():
            if req_id not in norms:
                seqs[req_id] = list(tables.keys())
                norms[req_id] = {}
            for seq_id, table in tables.items():
                if seq_id not in norms[req_id]:
                    # the reference seeds a new entry with a single 0.0 on the very first call
                    # ([0.0 * len]) and with len(table) zeros afterwards (:28 vs :44)
                    norms[req_id][seq_id] = [0.0] if first else [0.0] * len(table)
        for req_id in [r for r in norms if r not in reqid_mapping_blocktables]:
            del norms[req_id]
            del seqs[req_id]
        if changed:
            self.flat_sequence_map = [(r, s) for r, ss in seqs.items() for s in ss]
        return changed

    # -- lookups ---------------------------------------------------------------------------
    def get_reqid_and_seqid(self, seq_idx: int) -> Tuple[str, int]:
        assert seq_idx < len(self.flat_sequence_map), \
            f"seq_idx {seq_idx} is out of range length {len(self.flat_sequence_map)}"
        return self.flat_sequence_map[seq_idx]

    def get_req_block_tables(self, seq_idx: int) -> List[int]:
        req_id, seq_id = self.get_reqid_and_seqid(seq_idx)
        return self.reqid_mapping_blocktables[req_id][seq_id]

    # -- scores ----------------------------------------------------------------------------
    def update_block_l2norms(self, seq_idx: int, l2_norms: List[float]) -> None:
        req_id, seq_id = self.get_reqid_and_seqid(seq_idx)
        self.reqid_mapping_block_l2norms[req_id][seq_id] = l2_norms

    def update_last_block_l2norm(self, seq_idx: int, l2_norm: float) -> List[float]:
        req_id, seq_id = self.get_reqid_and_seqid(seq_idx)
        lst = self.reqid_mapping_block_l2norms[req_id][seq_id]
        lst.append(l2_norm)
        return lst

    def get_updated_block_tables(self, seq_idx: int, rmv_idx: int,
                                 is_rmv_l2norm: bool = False) -> List[int]:
        """Copy of the sequence's block table without entry rmv_idx; optionally also forget
        that block's score."""
        req_id, seq_id = self.get_reqid_and_seqid(seq_idx)
        table = list(self.reqid_mapping_blocktables[req_id][seq_id])
        del table[rmv_idx]
        if is_rmv_l2norm:
            del self.reqid_mapping_block_l2norms[req_id][seq_id][rmv_idx]
        return table
