// placeholder until the RIS restatement lands
