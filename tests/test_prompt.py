

def test_w2v_bert_layers_past_the_one_that_is_read_are_dropped_same_features():
    """`hidden_states[17]` is the input of encoder layer 17 (infer_v2.py:200-209 reads it from a 24-layer pass): `W2vBert` keeps
    layers 0..16 only -- the tensor is the same, bit for bit (small random-weight model of the installed transformers class)."""
    import copy

    import torch
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel

    import voice_tts_amd.prompt as PR

    torch.manual_seed(0)
    cfg = Wav2Vec2BertConfig(hidden_size=64, num_hidden_layers=6, num_attention_heads=4, intermediate_size=128, feature_projection_input_dim=160)
    m = Wav2Vec2BertModel(cfg).eval()
    audio = torch.randn(1, 16000)
    w = PR.W2vBert(copy.deepcopy(m), torch.zeros(64), torch.ones(64), layer=4)
    assert len(w.model.encoder.layers) == 4 and len(m.encoder.layers) == 6
    inp = w.extractor(audio, sampling_rate=16000, return_tensors="pt")
    with torch.no_grad():
        ref = m(input_features=inp["input_features"], attention_mask=inp["attention_mask"], output_hidden_states=True).hidden_states[4]
    assert torch.equal(w(audio), ref)
