"""MI355X-native streaming RNN-Transducer inference path (drop-in for the reference's
OnlineRNNTModel / online_rnnt_decode.py hot path).  See DESIGN.md."""
__all__ = ["testing"]
