"""Hot-path blocks with the reference's Block + ring interface."""
from .block_base import Block, COMMAND_OK, COMMAND_NOT_RECOGNIZED, COMMAND_WRONG_TYPE, COMMAND_INVALID
from .corr_block import Corr, regtile_index, tri_index
from .corr_acc_block import CorrAcc
from .beamform_block import Beamform
from .beamform_sum_beams_block import BeamformSumBeams
from .copy_block import Copy
from .corr_subsel_block import CorrSubsel
from .corr_output_full_block import CorrOutputFull
from .snap2_ingest_block import Snap2Ingest
from .beamform_output_block import BeamformOutput
from .corr_output_part_block import CorrOutputPart

__all__ = ["Block", "Corr", "CorrAcc", "Beamform", "BeamformSumBeams", "Copy", "CorrSubsel", "CorrOutputFull", "Snap2Ingest", "BeamformOutput", "CorrOutputPart", "regtile_index", "tri_index",
           "COMMAND_OK", "COMMAND_NOT_RECOGNIZED", "COMMAND_WRONG_TYPE", "COMMAND_INVALID"]
