def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **kw):
    return hidden_states, res_hidden_states
