def map_old_state_dict_weights(state_dict, mapping):
    for k in list(state_dict.keys()):
        for old, new in mapping.items():
            if old in k:
                state_dict[k.replace(old, new)] = state_dict.pop(k)
    return state_dict
