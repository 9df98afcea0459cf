"""vmrframe_amd -- MI355X-native (gfx950) implementation of the SeqPAN
cross-modal matching hot path of renjie-liang/VMRFrame.

Exports mirror what the reference's `main.py` resolves by name after
`from models import *` (reference main.py:21,87,99; utils/DataLoader.py:5-6).
"""
from .SeqPAN import (BaseFast, SeqPAN, infer_BaseFast, infer_basic, infer_basic_device, infer_SeqPAN,  # noqa: F401
                     lossfun_loc, lossfun_match, train_engine_BaseFast, train_engine_SeqPAN)
from .metrics import IoUMeter, append_ious, get_i345_mi  # noqa: F401
from .staging import FeatureArena  # noqa: F401
from .ban_map import ProposalMap2D, train_engine_ProposalMap2D  # noqa: F401  (row N2: the BAN 2-D proposal-map stage)
from .ban_encoders import QueryEncoder, VisualEncoder  # noqa: F401  (row N2: BAN's bi-LSTM encoders, reference models/BANlib/model.py:8-86)
from .ban import BAN, infer_BAN, train_engine_BAN  # noqa: F401  (row N2: the assembled model and its engine, reference models/BAN.py)

__all__ = ["SeqPAN", "train_engine_SeqPAN", "infer_SeqPAN", "BaseFast", "train_engine_BaseFast", "infer_BaseFast",
           "infer_basic", "infer_basic_device", "lossfun_loc", "lossfun_match", "IoUMeter", "append_ious", "get_i345_mi", "FeatureArena",
           "ProposalMap2D", "train_engine_ProposalMap2D", "VisualEncoder", "QueryEncoder", "BAN", "train_engine_BAN", "infer_BAN"]
