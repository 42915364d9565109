"""dump1090_amd - MI355X (gfx950) replacement for dump1090's IQ -> message hot path.

    Demodulator      GPU scan + demod (libmodes_gfx950.so) + in-order host resolve
                     (libmodes_host.so): bytes in, `struct modesMessage` list out.
    HostResolver     the host half alone (records -> messages); runs without a GPU.
    Tracker          aircraft table, CPR positions, SBS (port 30003) lines of a message list.
    distributed      one process per GPU: shard buffers, gather records, resolve on rank 0.

The package holds no CPU implementation of the GPU stages: constructing a Demodulator
without the HIP library / a GPU raises ModesError.
"""
from ._native import (BLOCK_POSITIONS, BLOCK_STRIDE, CARRY_BYTES, CARRY_SAMPLES, DATA_LEN, RECORD_DTYPE, ModesError,
                      ModesMessage)
from .demod import (Demodulator, HostResolver, Message, Tracker, block_count, onlyaddr_text, raw_net_text, raw_text,
                    shard_blocks, shard_byte_range, verbose_text)

__all__ = ["Demodulator", "HostResolver", "Tracker", "raw_net_text", "Message", "ModesError", "ModesMessage", "RECORD_DTYPE", "block_count",
           "shard_blocks", "shard_byte_range", "raw_text", "onlyaddr_text", "verbose_text", "DATA_LEN", "CARRY_BYTES",
           "BLOCK_STRIDE", "BLOCK_POSITIONS", "CARRY_SAMPLES"]
